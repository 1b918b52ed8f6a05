/*
 * rectdetect-mi355x: rectangle detector API of the reference (reference oclrect.h:5-23).
 * Device stages: rectdetect_amd/csrc/rd_rect.hip; host post-process: rectdetect_amd/csrc/rd_post.c.
 * Include vec234.h first (rect_t is built from vec2 / vec3).
 */
#ifndef RD_COMPAT_OCLRECT_H
#define RD_COMPAT_OCLRECT_H
#include <stdint.h>
#if defined(__cplusplus)
extern "C" {
#endif

/* Element 0 of a returned array carries nItems (count INCLUDING element 0); elements 1.. are rectangles:
 * image corners c2, 3-D corners c3, residual of the pose fit, status (bit 0: looks like a screen, bit 1: found by the
 * per-polyline pass).  (reference oclrect.h:5-15) */
struct oclimgutil_t;
struct oclpolyline_t;
struct oclrect_t;

typedef struct rect_t {
  union {
    struct {
      vec2 c2[4];
      vec3 c3[4];
      double value;
      uint32_t status;
    };
    int nItems;
  };
} rect_t;

struct oclrect_t *init_oclrect(struct oclimgutil_t *oclimgutil, struct oclpolyline_t *oclpolyline, cl_device_id device, cl_context context, cl_command_queue queue, int iw, int ih);
void dispose_oclrect(struct oclrect_t *thiz);

/* synchronous: BGR u8 frame with row stride ws (ws*ih <= 4*iw*ih) -> malloc'd rect_t array owned by the caller
 * (reference oclrect.c:1230-1246) */
rect_t *oclrect_executeOnce(struct oclrect_t *thiz, uint8_t *imgData, int ws, const double tanAOV);

/* two-deep pipeline: strictly alternate enqueue / poll after the first enqueue (reference oclrect.c:1248-1278) */
void oclrect_enqueueTask(struct oclrect_t *thiz, uint8_t *imgData, int ws);
rect_t *oclrect_pollTask(struct oclrect_t *thiz, const double tanAOV);

#if defined(__cplusplus)
}
#endif
#endif
