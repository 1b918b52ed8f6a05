/*
 * rectdetect-mi355x: polyline extraction API of the reference (reference oclpolyline.h:74-88), implemented in
 * rectdetect_amd/csrc/rd_k_poly.hip on compacted edge pixels.
 */
#ifndef RD_COMPAT_OCLPOLYLINE_H
#define RD_COMPAT_OCLPOLYLINE_H
#include <stdint.h>
#if defined(__cplusplus)
extern "C" {
#endif

/* reference oclpolyline.h:5-72 lists one cl_kernel per OpenCL kernel here; nothing outside reads them */
typedef struct oclpolyline_t {
  uint32_t magic;
  cl_device_id device;
  cl_context context;
  void *impl;
} oclpolyline_t;

/* One output record, 56 bytes (reference oclpolyline.h:74-83).  Record 0 of a list is a header whose first int is
 * the number n of records; records 1..n are valid when polyid != 0 and are chained by leftPtr / rightPtr (0 = none). */
typedef struct linesegment_t {
  float x0, y0, x1, y1;
  int32_t startIndex, endIndex;
  int32_t leftPtr, rightPtr;
  int32_t startCount, endCount;
  int32_t maxDist;
  int32_t polyid;
  int32_t npix;
  int32_t level;
} linesegment_t;

oclpolyline_t *init_oclpolyline(cl_device_id device, cl_context context);
void dispose_oclpolyline(oclpolyline_t *thiz);

/* reference oclpolyline.c:218-309.  lsList: lsListSize bytes of linesegment_t (output); lsIdOut: per-pixel segment id
 * (output); in: 0/1 edge mask (int plane, read only); tmp0: 16*iw*ih-byte scratch; tmp1..tmp6: 4*iw*ih-byte scratch.
 * The 2-pixel frame ring of tmp3 (the reference's gap-bridging output plane) is READ: the reference's kernel leaves
 * that ring unwritten, so its previous content takes part in the result. */
cl_event oclpolyline_execute(oclpolyline_t *thiz, cl_mem lsList, int lsListSize, cl_mem lsIdOut, cl_mem in, cl_mem tmp0, cl_mem tmp1, cl_mem tmp2, cl_mem tmp3, cl_mem tmp4, cl_mem tmp5, cl_mem tmp6, float minerror, int sizeThre, int iw, int ih, cl_command_queue queue, const cl_event *events);

#if defined(__cplusplus)
}
#endif
#endif
