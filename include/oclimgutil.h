/*
 * rectdetect-mi355x: image-operator API of the reference (reference oclimgutil.h:74-100), implemented as
 * hand-written gfx950 HIP kernels in rectdetect_amd/csrc/rd_imgutil.hip.
 *
 * Conventions (identical to the reference):
 *  - planes are row-major, iw elements per row, 4 bytes per pixel unless noted; "plab" = packed Lab uint32
 *    (bits 0-11 L, 12-21 a, 22-31 b); float2 planes are 8 bytes per pixel; BGR images are u8 with row stride ws
 *  - oclimgutil_clear/copy/rand take sizes in BYTES, cast/threshold take element counts
 *  - every op is enqueued on `queue` (a HIP stream) and returns NULL when `events` is NULL, otherwise a new
 *    event the caller releases; `events` is a NULL-terminated list to wait for
 *  - naming is convert_<out>_<in>: oclimgutil_convert_plab_bgr converts BGR to plab
 *  - errors are fatal (message on stderr, exit(-1))
 */
#ifndef RD_COMPAT_OCLIMGUTIL_H
#define RD_COMPAT_OCLIMGUTIL_H
#include <stdint.h>
#if defined(__cplusplus)
extern "C" {
#endif

/* The reference exposes one cl_kernel + kernel id per OpenCL kernel in this struct (oclimgutil.h:5-72); no
 * caller reads them.  Kernels are compiled ahead of time here, so only the leading fields remain. */
typedef struct oclimgutil_t {
  uint32_t magic;
  cl_device_id device;
  cl_context context;
  void *impl;
} oclimgutil_t;

oclimgutil_t *init_oclimgutil(cl_device_id device, cl_context context);   /* reference oclimgutil.c:20-102 */
void dispose_oclimgutil(oclimgutil_t *thiz);                              /* :104-135 */

cl_event oclimgutil_clear(oclimgutil_t *thiz, cl_mem out, int size, cl_command_queue queue, const cl_event *events);
cl_event oclimgutil_copy(oclimgutil_t *thiz, cl_mem out, cl_mem in, int size, cl_command_queue queue, const cl_event *events);
cl_event oclimgutil_cast_i_f(oclimgutil_t *thiz, cl_mem out, cl_mem in, float scale, int size, cl_command_queue queue, const cl_event *events);
cl_event oclimgutil_cast_c_i(oclimgutil_t *thiz, cl_mem out, cl_mem in, int size, cl_command_queue queue, const cl_event *events);
cl_event oclimgutil_threshold_i_i(oclimgutil_t *thiz, cl_mem out, cl_mem in, int vlow, int threshold, int vhigh, int size, cl_command_queue queue, const cl_event *events);
cl_event oclimgutil_threshold_f_f(oclimgutil_t *thiz, cl_mem out, cl_mem in, float vlow, float threshold, float vhigh, int size, cl_command_queue queue, const cl_event *event);
cl_event oclimgutil_rand(oclimgutil_t *thiz, cl_mem out, int size, cl_command_queue queue, const cl_event *events);
cl_event oclimgutil_convert_bgr_luminancef(oclimgutil_t *thiz, cl_mem out, cl_mem in, int iw, int ih, int ws, cl_command_queue queue, const cl_event *events);
cl_event oclimgutil_convert_bgr_lumaf(oclimgutil_t *thiz, cl_mem out, cl_mem in, float f, int iw, int ih, int ws, cl_command_queue queue, const cl_event *events);
cl_event oclimgutil_convert_bgr_labeli(oclimgutil_t *thiz, cl_mem out, cl_mem in, int bgc, int iw, int ih, int ws, cl_command_queue queue, const cl_event *events);
cl_event oclimgutil_edge_f_f(oclimgutil_t *thiz, cl_mem out, cl_mem in, int iw, int ih, cl_command_queue queue, const cl_event *events);
cl_event oclimgutil_edgevec_f2_f(oclimgutil_t *thiz, cl_mem out, cl_mem in, int iw, int ih, cl_command_queue queue, const cl_event *events);
cl_event oclimgutil_thinthres_f_f_f2(oclimgutil_t *thiz, cl_mem out, cl_mem in, cl_mem vec, int iw, int ih, cl_command_queue queue, const cl_event *events);
cl_event oclimgutil_thincubic_f_f_f2(oclimgutil_t *thiz, cl_mem out, cl_mem in, cl_mem vec, int iw, int ih, cl_command_queue queue, const cl_event *events);
cl_event oclimgutil_label8x_int_int(oclimgutil_t *thiz, cl_mem out, cl_mem in, cl_mem tmp, int bgc, int iw, int ih, cl_command_queue queue, const cl_event *events);
cl_event oclimgutil_iirblur_f_f(oclimgutil_t *thiz, cl_mem obuf, cl_mem ibuf, cl_mem tmp0, cl_mem tmp1, int r, int iw, int ih, cl_command_queue queue, const cl_event *events);
cl_event oclimgutil_convert_plab_bgr(oclimgutil_t *thiz, cl_mem out, cl_mem in, int iw, int ih, int ws, cl_command_queue queue, const cl_event *events);
cl_event oclimgutil_convert_bgr_plab(oclimgutil_t *thiz, cl_mem out, cl_mem in, int iw, int ih, int ws, cl_command_queue queue, const cl_event *events);
cl_event oclimgutil_unpack_f_f_f_plab(oclimgutil_t *thiz, cl_mem out0, cl_mem out1, cl_mem out2, cl_mem in, int iw, int ih, cl_command_queue queue, const cl_event *events);
cl_event oclimgutil_pack_plab_f_f_f(oclimgutil_t *thiz, cl_mem out, cl_mem in0, cl_mem in1, cl_mem in2, int iw, int ih, cl_command_queue queue, const cl_event *events);
cl_event oclimgutil_edgevec_f2_plab(oclimgutil_t *thiz, cl_mem out, cl_mem in, int iw, int ih, cl_command_queue queue, const cl_event *events);
cl_event oclimgutil_edge_f_plab(oclimgutil_t *thiz, cl_mem out, cl_mem in, int iw, int ih, cl_command_queue queue, const cl_event *events);
cl_event oclimgutil_calcStrength(oclimgutil_t *thiz, cl_mem out, cl_mem edge, cl_mem label, int iw, int ih, cl_command_queue queue, const cl_event *events);
cl_event oclimgutil_filterStrength(oclimgutil_t *thiz, cl_mem labelinout, cl_mem str, int thre, int iw, int ih, cl_command_queue queue, const cl_event *events);

#if defined(__cplusplus)
}
#endif
#endif
