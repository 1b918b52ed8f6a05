// rectdetect-mi355x: the reference's public operator API (oclimgutil.h, oclpolyline.h, oclrect.h) and the rd_detector
// extension on top of the gfx950 kernels.  C ABI, opaque handles, fatal-on-error like the reference.
#include "rd_internal.h"
#include "rd_kernels.h"
#include "rd_poly_scratch.h"
#include "rectdetect_hip.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <time.h>
#include <sched.h>

extern "C" {
#include "vec234.h"
#include "oclimgutil.h"
#include "oclpolyline.h"
#include "oclrect.h"
#include "rd_post.h"
}

using rdrt::dptr;
using rdrt::stream;

// Environment switches.  A detector reads the ones include/rectdetect_hip.h lists ("Environment") when it is created - its behaviour never changes afterwards, and two
// detectors of one process may differ.  The switches of experiments whose variant was measured and NOT kept (profiles/NOTES_r0*.md) exist in tuning builds only
// (-DRD_TUNING, tools/variants.sh): in the product they are the constants below.
static const char *rd_env(const char *name) { const char *v = getenv(name); return (v && *v) ? v : NULL; }
static int rd_env_int(const char *name, int dflt) { const char *v = rd_env(name); return v ? atoi(v) : dflt; }
#ifdef RD_TUNING
#define RD_LAB_INT(name, dflt) rd_env_int(name, dflt)
#else
#define RD_LAB_INT(name, dflt) (dflt)
#endif

#define MAGIC_IMGUTIL 0xa640d893u
#define MAGIC_POLYLINE 0x808f3801u
#define MAGIC_RECT 0x808f3802u

template <typename T> static T *dnew(size_t n) { void *p = NULL; RD_HIP(hipMalloc(&p, (n ? n : 1) * sizeof(T))); return (T *)p; }
static void dfree(void *p) { if (p) RD_HIP(hipFree(p)); }

// ================================================================================================ oclimgutil
struct ImgutilImpl { int N; size_t ntails; float *s[3]; float *tails; int *flags; };

static void imgutil_scratch(oclimgutil_t *t, int iw, int ih) {
  ImgutilImpl *im = (ImgutilImpl *)t->impl;
  const int N = iw * ih;
  const size_t a = rdk::iir_pass_scratch_floats(1, ih, iw), b = rdk::iir_pass_scratch_floats(1, iw, ih), nt = a > b ? a : b;
  if (im->N >= N && im->ntails >= nt) return;
  for (int k = 0; k < 3; k++) { dfree(im->s[k]); im->s[k] = dnew<float>((size_t)N); }
  dfree(im->tails); dfree(im->flags);
  im->tails = dnew<float>(nt);
  im->flags = dnew<int>(16); RD_HIP(hipMemset(im->flags, 0, 16 * sizeof(int))); RD_HIP(hipStreamSynchronize(0));
  im->N = N; im->ntails = nt;
}

extern "C" {

oclimgutil_t *init_oclimgutil(cl_device_id device, cl_context context) {
  oclimgutil_t *t = (oclimgutil_t *)calloc(1, sizeof(*t));
  t->magic = MAGIC_IMGUTIL; t->device = device; t->context = context;
  t->impl = calloc(1, sizeof(ImgutilImpl));
  if (device) RD_HIP(hipSetDevice(device->ordinal));
  return t;
}

void dispose_oclimgutil(oclimgutil_t *t) {
  if (!t || t->magic != MAGIC_IMGUTIL) exitf(-1, "dispose_oclimgutil: bad handle\n");
  ImgutilImpl *im = (ImgutilImpl *)t->impl;
  for (int k = 0; k < 3; k++) dfree(im->s[k]);
  dfree(im->tails); dfree(im->flags);
  free(im);
  t->magic = 0;
  free(t);
}

#define IU_BEGIN(name)                                                                   \
  if (!thiz || thiz->magic != MAGIC_IMGUTIL) exitf(-1, name ": bad oclimgutil handle\n"); \
  rdrt::wait_list(queue, events);                                                        \
  hipStream_t s = stream(queue)
#define IU_END(name) rdrt::check_launch(name); return rdrt::finish_op(queue, events)

cl_event oclimgutil_clear(oclimgutil_t *thiz, cl_mem out, int size, cl_command_queue queue, const cl_event *events) {
  IU_BEGIN("oclimgutil_clear");
  rdk::clear_i(s, (int *)dptr(out), (size + 3) / 4);
  IU_END("oclimgutil_clear");
}
cl_event oclimgutil_copy(oclimgutil_t *thiz, cl_mem out, cl_mem in, int size, cl_command_queue queue, const cl_event *events) {
  IU_BEGIN("oclimgutil_copy");
  rdk::copy_i(s, (int *)dptr(out), (const int *)dptr(in), (size + 3) / 4);
  IU_END("oclimgutil_copy");
}
cl_event oclimgutil_cast_i_f(oclimgutil_t *thiz, cl_mem out, cl_mem in, float scale, int size, cl_command_queue queue, const cl_event *events) {
  IU_BEGIN("oclimgutil_cast_i_f");
  rdk::cast_i_f(s, (int *)dptr(out), (const float *)dptr(in), scale, size);
  IU_END("oclimgutil_cast_i_f");
}
cl_event oclimgutil_cast_c_i(oclimgutil_t *thiz, cl_mem out, cl_mem in, int size, cl_command_queue queue, const cl_event *events) {
  IU_BEGIN("oclimgutil_cast_c_i");
  rdk::cast_c_i(s, (int8_t *)dptr(out), (const int *)dptr(in), size);
  IU_END("oclimgutil_cast_c_i");
}
cl_event oclimgutil_threshold_i_i(oclimgutil_t *thiz, cl_mem out, cl_mem in, int vlow, int threshold, int vhigh, int size, cl_command_queue queue, const cl_event *events) {
  IU_BEGIN("oclimgutil_threshold_i_i");
  rdk::threshold_i(s, (int *)dptr(out), (const int *)dptr(in), vlow, threshold, vhigh, size);
  IU_END("oclimgutil_threshold_i_i");
}
cl_event oclimgutil_threshold_f_f(oclimgutil_t *thiz, cl_mem out, cl_mem in, float vlow, float threshold, float vhigh, int size, cl_command_queue queue, const cl_event *events) {
  IU_BEGIN("oclimgutil_threshold_f_f");
  rdk::threshold_f(s, (float *)dptr(out), (const float *)dptr(in), vlow, threshold, vhigh, size);
  IU_END("oclimgutil_threshold_f_f");
}
cl_event oclimgutil_rand(oclimgutil_t *thiz, cl_mem out, int size, cl_command_queue queue, const cl_event *events) {
  IU_BEGIN("oclimgutil_rand");
  rdk::rand_i(s, (int *)dptr(out), 0, (size + 3) / 4);   // the reference's wrapper never sets the seed argument (oclimgutil.c:178-183)
  IU_END("oclimgutil_rand");
}
cl_event oclimgutil_convert_plab_bgr(oclimgutil_t *thiz, cl_mem out, cl_mem in, int iw, int ih, int ws, cl_command_queue queue, const cl_event *events) {
  IU_BEGIN("oclimgutil_convert_plab_bgr");
  rdk::bgr2plab(s, (uint32_t *)dptr(out), (const uint8_t *)dptr(in), iw, ih, ws);
  IU_END("oclimgutil_convert_plab_bgr");
}
cl_event oclimgutil_unpack_f_f_f_plab(oclimgutil_t *thiz, cl_mem out0, cl_mem out1, cl_mem out2, cl_mem in, int iw, int ih, cl_command_queue queue, const cl_event *events) {
  IU_BEGIN("oclimgutil_unpack_f_f_f_plab");
  rdk::unpack_plab(s, (float *)dptr(out0), (float *)dptr(out1), (float *)dptr(out2), (const uint32_t *)dptr(in), iw * ih);
  IU_END("oclimgutil_unpack_f_f_f_plab");
}
cl_event oclimgutil_pack_plab_f_f_f(oclimgutil_t *thiz, cl_mem out, cl_mem in0, cl_mem in1, cl_mem in2, int iw, int ih, cl_command_queue queue, const cl_event *events) {
  IU_BEGIN("oclimgutil_pack_plab_f_f_f");
  rdk::pack_plab(s, (uint32_t *)dptr(out), (const float *)dptr(in0), (const float *)dptr(in1), (const float *)dptr(in2), iw * ih);
  IU_END("oclimgutil_pack_plab_f_f_f");
}

// oclimgutil.c:248-273.  obuf = vertical(horizontal(ibuf)); tmp0 / tmp1 are scratch.  r = 2 (sigma 1, the only radius any
// caller in the reference passes) takes the blocked evaluation of the frame path (scratch only written when that fails its
// on-device check); every other radius of the table (0..31, sigma = (r + 1) / 3) the full-length sweeps.
cl_event oclimgutil_iirblur_f_f(oclimgutil_t *thiz, cl_mem obuf, cl_mem ibuf, cl_mem tmp0, cl_mem tmp1, int r, int iw, int ih, cl_command_queue queue, const cl_event *events) {
  IU_BEGIN("oclimgutil_iirblur_f_f");
  if (r < 0 || r > RD_IIR_MAX_R) exitf(-1, "oclimgutil_iirblur_f_f: r = %d is outside the coefficient table (0..%d; the reference reads beyond iircoef[] there)\n", r, RD_IIR_MAX_R);
  imgutil_scratch(thiz, iw, ih);
  ImgutilImpl *im = (ImgutilImpl *)thiz->impl;
  float *o = (float *)dptr(obuf), *t0 = (float *)dptr(tmp0), *t1 = (float *)dptr(tmp1);
  const float *in = (const float *)dptr(ibuf);
  float *d1[3] = { im->s[0], NULL, NULL }; const float *s1[3] = { in, NULL, NULL };
  if (r != 2) {
    // (the mirrored run-in of r + 9 samples must stay inside the line: iu:551 reads mirror1(x) unchecked)
    if (iw < r + 11 || ih < r + 11) exitf(-1, "oclimgutil_iirblur_f_f: r = %d needs planes of at least %d x %d (the reference reads outside a smaller one), got %d x %d\n", r, r + 11, r + 11, iw, ih);
    rdk::transpose_f(s, d1, s1, 1, iw, ih);                                 // s0 = in^T (ih wide)
    rdk::iir_blur_lines(s, t0, im->s[0], t0, t1, ih, iw, r);                // along x; t0 = horizontal result, transposed
    float *d2[3] = { im->s[1], NULL, NULL }; const float *s2[3] = { t0, NULL, NULL };
    rdk::transpose_f(s, d2, s2, 1, ih, iw);                                 // s1 = horizontal result, original layout
    rdk::iir_blur_lines(s, o, im->s[1], t0, t1, iw, ih, r);                 // along y
    IU_END("oclimgutil_iirblur_f_f");
  }
  rdk::transpose_f(s, d1, s1, 1, iw, ih);                                   // s0 = in^T (ih wide)
  float *fw[3] = { t0, NULL, NULL }, *bw[3] = { t1, NULL, NULL };
  { float *dst[3] = { im->s[1], NULL, NULL }; const float *src[3] = { im->s[0], NULL, NULL };
    rdk::iir_blur_pass(s, dst, src, fw, bw, 1, ih, iw, 1, im->tails, im->flags); }       // along x; s1 = horizontal result, original layout
  { float *dst[3] = { o, NULL, NULL }; const float *src[3] = { im->s[1], NULL, NULL };
    rdk::iir_blur_pass(s, dst, src, fw, bw, 1, iw, ih, 0, im->tails, im->flags + 1); }   // along y
  IU_END("oclimgutil_iirblur_f_f");
}

cl_event oclimgutil_edgevec_f2_f(oclimgutil_t *thiz, cl_mem out, cl_mem in, int iw, int ih, cl_command_queue queue, const cl_event *events) {
  IU_BEGIN("oclimgutil_edgevec_f2_f");
  rdk::edgevec(s, (float *)dptr(out), (const float *)dptr(in), iw, ih);
  IU_END("oclimgutil_edgevec_f2_f");
}
cl_event oclimgutil_edge_f_plab(oclimgutil_t *thiz, cl_mem out, cl_mem in, int iw, int ih, cl_command_queue queue, const cl_event *events) {
  IU_BEGIN("oclimgutil_edge_f_plab");
  rdk::edge_plab(s, (float *)dptr(out), (const uint32_t *)dptr(in), iw, ih);
  IU_END("oclimgutil_edge_f_plab");
}
cl_event oclimgutil_thinthres_f_f_f2(oclimgutil_t *thiz, cl_mem out, cl_mem in, cl_mem vec, int iw, int ih, cl_command_queue queue, const cl_event *events) {
  IU_BEGIN("oclimgutil_thinthres_f_f_f2");
  rdk::thinthres(s, (float *)dptr(out), (const float *)dptr(in), (const float *)dptr(vec), iw, ih);
  IU_END("oclimgutil_thinthres_f_f_f2");
}
// oclimgutil.c:227-246: converged labelling instead of 1 + 10 propagation passes; `tmp` (the reference's pass flags) is unused
cl_event oclimgutil_label8x_int_int(oclimgutil_t *thiz, cl_mem out, cl_mem in, cl_mem tmp, int bgc, int iw, int ih, cl_command_queue queue, const cl_event *events) {
  IU_BEGIN("oclimgutil_label8x_int_int");
  (void)tmp;
  rdk::label8(s, (int *)dptr(out), (const int *)dptr(in), bgc, iw, ih);
  IU_END("oclimgutil_label8x_int_int");
}
cl_event oclimgutil_calcStrength(oclimgutil_t *thiz, cl_mem out, cl_mem edge, cl_mem label, int iw, int ih, cl_command_queue queue, const cl_event *events) {
  IU_BEGIN("oclimgutil_calcStrength");
  rdk::calc_strength(s, (int *)dptr(out), (const float *)dptr(edge), (int *)dptr(label), iw, ih);
  IU_END("oclimgutil_calcStrength");
}
cl_event oclimgutil_filterStrength(oclimgutil_t *thiz, cl_mem labelinout, cl_mem str, int thre, int iw, int ih, cl_command_queue queue, const cl_event *events) {
  IU_BEGIN("oclimgutil_filterStrength");
  rdk::filter_strength(s, (int *)dptr(labelinout), (const int *)dptr(str), thre, iw, ih);
  IU_END("oclimgutil_filterStrength");
}

// Debug visualisers and operators no application calls (SURVEY.md 8a "dead for the apps", 8f rank 4).
cl_event oclimgutil_convert_bgr_lumaf(oclimgutil_t *thiz, cl_mem out, cl_mem in, float f, int iw, int ih, int ws, cl_command_queue queue, const cl_event *events) {
  IU_BEGIN("oclimgutil_convert_bgr_lumaf");
  rdk::convert_bgr_lumaf(s, (uint8_t *)dptr(out), (const float *)dptr(in), f, iw, ih, ws);
  IU_END("oclimgutil_convert_bgr_lumaf");
}
cl_event oclimgutil_convert_bgr_labeli(oclimgutil_t *thiz, cl_mem out, cl_mem in, int bgc, int iw, int ih, int ws, cl_command_queue queue, const cl_event *events) {
  IU_BEGIN("oclimgutil_convert_bgr_labeli");
  rdk::convert_bgr_labeli(s, (uint8_t *)dptr(out), (const int *)dptr(in), bgc, iw, ih, ws);
  IU_END("oclimgutil_convert_bgr_labeli");
}
cl_event oclimgutil_convert_bgr_plab(oclimgutil_t *thiz, cl_mem out, cl_mem in, int iw, int ih, int ws, cl_command_queue queue, const cl_event *events) {
  IU_BEGIN("oclimgutil_convert_bgr_plab");
  rdk::plab2bgr(s, (uint8_t *)dptr(out), (const uint32_t *)dptr(in), iw, ih, ws);
  IU_END("oclimgutil_convert_bgr_plab");
}
cl_event oclimgutil_edge_f_f(oclimgutil_t *thiz, cl_mem out, cl_mem in, int iw, int ih, cl_command_queue queue, const cl_event *events) {
  IU_BEGIN("oclimgutil_edge_f_f");
  rdk::edge_f(s, (float *)dptr(out), (const float *)dptr(in), iw, ih);
  IU_END("oclimgutil_edge_f_f");
}
cl_event oclimgutil_edgevec_f2_plab(oclimgutil_t *thiz, cl_mem out, cl_mem in, int iw, int ih, cl_command_queue queue, const cl_event *events) {
  IU_BEGIN("oclimgutil_edgevec_f2_plab");
  rdk::edgevec_plab(s, (float *)dptr(out), (const uint32_t *)dptr(in), iw, ih);
  IU_END("oclimgutil_edgevec_f2_plab");
}
cl_event oclimgutil_thincubic_f_f_f2(oclimgutil_t *thiz, cl_mem out, cl_mem in, cl_mem vec, int iw, int ih, cl_command_queue queue, const cl_event *events) {
  IU_BEGIN("oclimgutil_thincubic_f_f_f2");
  rdk::thincubic(s, (float *)dptr(out), (const float *)dptr(in), (const float *)dptr(vec), iw, ih);
  IU_END("oclimgutil_thincubic_f_f_f2");
}
// The reference sets 5 of the 6 arguments of its convert_bgr_luminancef kernel (oclimgutil.c:187 "MMiii" against
// iu:275: out, in, f, iw, ih, ws), so the call cannot succeed on a conforming OpenCL runtime (CL_INVALID_KERNEL_ARGS ->
// exitf); it fails here in the same way, with the reason spelled out.
cl_event oclimgutil_convert_bgr_luminancef(oclimgutil_t *, cl_mem, cl_mem, int, int, int, cl_command_queue, const cl_event *) {
  exitf(-1, "oclimgutil_convert_bgr_luminancef: the reference passes 5 of this kernel's 6 arguments (oclimgutil.c:187); the call is an error there and here\n");
  return NULL;
}

}  // extern "C"

// ================================================================================================ oclpolyline
struct PolylineImpl { int iw, ih; rdk::PolyScratch *ps; };

extern "C" {

oclpolyline_t *init_oclpolyline(cl_device_id device, cl_context context) {
  oclpolyline_t *t = (oclpolyline_t *)calloc(1, sizeof(*t));
  t->magic = MAGIC_POLYLINE; t->device = device; t->context = context;
  t->impl = calloc(1, sizeof(PolylineImpl));
  if (device) RD_HIP(hipSetDevice(device->ordinal));
  return t;
}

void dispose_oclpolyline(oclpolyline_t *t) {
  if (!t || t->magic != MAGIC_POLYLINE) exitf(-1, "dispose_oclpolyline: bad handle\n");
  PolylineImpl *im = (PolylineImpl *)t->impl;
  rdk::poly_scratch_destroy(im->ps);
  free(im);
  t->magic = 0;
  free(t);
}

// oclpolyline.c:218-309.  The caller's scratch planes are not needed by this implementation (it keeps its own compact
// scratch), except that the ring of tmp3 is read for the stale-ring semantics of the reference (see oclpolyline.h).
cl_event oclpolyline_execute(oclpolyline_t *thiz, cl_mem lsList, int lsListSize, cl_mem lsIdOut, cl_mem in, cl_mem tmp0, cl_mem tmp1, cl_mem tmp2, cl_mem tmp3,
                             cl_mem tmp4, cl_mem tmp5, cl_mem tmp6, float minerror, int sizeThre, int iw, int ih, cl_command_queue queue, const cl_event *events) {
  if (!thiz || thiz->magic != MAGIC_POLYLINE) exitf(-1, "oclpolyline_execute: bad handle\n");
  (void)tmp0; (void)tmp1; (void)tmp2; (void)tmp4; (void)tmp5; (void)tmp6;
  PolylineImpl *im = (PolylineImpl *)thiz->impl;
  if (!im->ps || im->iw != iw || im->ih != ih) {
    rdk::poly_scratch_destroy(im->ps);
    im->ps = rdk::poly_scratch_create(iw, ih);
    im->iw = iw; im->ih = ih;
  }
  rdrt::wait_list(queue, events);
  // the frame descriptor of this call (the kernels of the stage take their pointers from it: rd_poly_scratch.h)
  rdk::PolyFrame fr;
  memset(&fr, 0, sizeof(fr));
  fr.ps = *im->ps; fr.in = (const int *)dptr(in); fr.ring_src = (const int *)dptr(tmp3); fr.lslist = dptr(lsList); fr.ids = (int *)dptr(lsIdOut);
  rdk::polyline(stream(queue), &fr, 1, lsListSize, 0, minerror, sizeThre, iw, ih, 0);
  rdk::polyline_ids(stream(queue), &fr, 1, iw * ih);
  rdrt::check_launch("oclpolyline_execute");
  return rdrt::finish_op(queue, events);
}

}  // extern "C"

// ================================================================================================ detector
#define RD_NBUDGETS 7       // launch budgets of the region merge: 8, 10, .. 20 (kRoundBudgets)
#define RD_MAXREC 512       // records copied back per frame without a second transfer (the copy runs at PCIe speed: 16 us for 2048)

struct Slot {
  hipStream_t st;
  hipStream_t st2;                        // second stream of the slot: polyline stage, parallel to the region stages
  int uploaded_early;                     // group mode, host frames: the upload was issued when the frame was handed over (ev_fork marks its end)
  int pooled_streams;                     // st / st2 come from the process-wide pool of high-priority streams and go back there
  int shares_streams;                     // st / st2 belong to another slot (see rd_detector_create)
  hipEvent_t ev_fork, ev_mm, ev_join;
  hipEvent_t ev_begin, ev_done, ev_strong;   // ev_begin/ev_done carry timestamps: device time of the frame (rd_detector_counter)
  hipEvent_t watch_begin, watch_done;        // the events that bracket the frame in flight: the slot's own, or - a frame of a group launch - the ones of the group's first slot
  hipEvent_t ev_redo;                        // end of a repeated part of the frame (slot_finish_device)
  hipEvent_t ev_upload;                      // one or two frames in flight: the copy engine has read a host frame it took straight from the caller's pinned buffer
  hipStream_t st_redo;                       // created on first use: the slow absorption path (frame_absorb_slow), fetches of long lists
  int *big_probes; int big_probes_cap;       // probes of a frame with more segments than `probes` holds (grows on demand)
  const int8_t *prev_in;                  // the strong mask this slot's frame read (plane of rd_detector::prev_ring)
  uint8_t *bgr;
  const uint8_t *src;                     // where the frame in flight is read from: bgr (uploaded) or the caller's device buffer
  uint32_t *plab0, *plab1, *smooth, *quant;
  float *tr[3], *fw[3], *bw[3], *hz[3], *bl[3], *vxy, *strength, *nms;
  int *i0, *i1, *mask0, *tidy, *label1, *strsum, *region, *rsize, *scratch2, *d2s, *boundarysrc, *boundary, *lsid, *table, *claim, *probes, *region0, *tlist;
  int8_t *e8;
  unsigned long long *mmbits;             // the merge mask (oclrect.c:315-321) as a bit plane
  unsigned long long *strongbits;         // this frame's strong mask as a bit plane (ceil(iw / 64) words per row): what the polyline stage traces
  uint16_t *ext;
  float *tails; int *flags; int iir_chunked;
  void *lslist;
  rdk::PolyScratch *ps;
  rdk::PolyFrame *frame;                  // this slot's descriptor for the sparse stages (element `slot index` of the detector's array)
  hipEvent_t ev_dense;                    // batched mode: the frame's dense stages are done (the batch's sparse stages wait for it)
  int pending_sparse;                     // batched mode: dense stages enqueued, sparse stages not launched yet
  int pending_dense;                      // group mode (rd_detector::zb > 1): frame handed over, nothing launched yet
  int group_n;                            // frames that ran between this frame's ev_begin and ev_done (1, or the group's size: the interval is shared)
  hipGraphExec_t gz0, gz2[3 * RD_NBUDGETS]; int gz_ws;      // group mode, first slot of a group: the group's launch sequences (dense stages up to the first labelling; everything after the strong masks)
  // rectangles on the device (RD_DEVICE_POST): scratch, result block in pinned host memory, whether this frame's block is valid and for which aperture
  int *post_scratch, *h_post, *h_post_dev;
  int post_mode; double post_tan;
  // host side
  void *h_bgr;            // pinned staging for host frames
  void *h_segs; int *h_probes; int *h_ctr;   // views into h_pack
  int *h_pack, *h_pack_dev;                // everything the host needs from a frame, assembled by the device in pinned host memory (host / device address)
  long seq;
  int ws;
  // captured launch sequences (three segments, see enqueue_frame) and the stride they were captured for
  hipGraphExec_t gexec[3]; int graph_ws;   // gexec[2] unused: the last segment has one graph per round budget (gexec2)
  hipGraphExec_t gexec2[3 * RD_NBUDGETS];  // [round budget index][polyline mode]
  int poly_mode;                           // polyline mode of the frame in flight (1 = single block, 0 = multi-launch)
  int rounds;                             // region-merge round budget of the frame in flight
  // post-process worker
  pthread_t th; pthread_mutex_t mu; pthread_cond_t cv;
  int state;              // 0 idle, 1 submitted to the GPU, 2 result ready
  int quit;
  void *result; double result_tan; void *res_segs; int res_nsegs;
  struct rd_detector *owner;
};

struct rd_detector {
  uint32_t magic;
  int device, iw, ih, N, nslots, nworkers, maxrec_dev;
  // Sparse stages (polylines, votes, probes) of `batch` consecutive frame slots run as ONE set of launches (frame = blockIdx.z): they
  // are a chain of 20-odd latency-bound launches that keeps a hardware queue busy for ~0.4 ms without filling the chip, per
  // batch instead of per frame.  Slots [g * batch, (g + 1) * batch) form group g; 1 = every frame on its own (shortest latency).
  int batch; unsigned sparse_rot;
  int device_post;                        // candidate funnel + pose estimation on the device (rd_k_post.hip) instead of the host worker threads
  long n_post_device, n_post_host, host_post_ns;
  int defer, deferred_slot;               // batched mode: a complete group's sparse stages are launched only once the NEXT group's dense stages are enqueued (deferred_slot: a slot of the waiting group or -1)
  rdk::PolyFrame *frames;                 // nslots descriptors (host memory; they travel as kernel arguments), slot order
  Slot *slots;
  int nstreams;
  int zb;                 // frames per launch of the DENSE stages as well (groups of zb consecutive slots, frame = blockIdx.z): small frames, whose launches do not fill the device
  char *arena; size_t slot_pitch;      // zb > 1: all slots' planes in one allocation, slot k at arena + k * slot_pitch
  // strong-edge masks of the last frames (reference quirk H1: a frame's strength sums start from the mask of the frame before), one byte per
  // pixel: a ring of nslots + 1 planes - frame t reads plane t mod (nslots + 1) and writes the next one, so what a frame read stays intact as
  // long as its slot's planes do (debug plane "strsum")
  int8_t *prev_ring; int nring;
  int t_edge, t_strong;                   // thresholds of the strength sums (500, 2500; RD_TEST_THRESHOLDS)
  int strong_by_frame;                    // tests (RD_STRONG_BY_FRAME): the strong masks of a group frame by frame instead of in one launch
  long n_strong_group, n_strong_by_frame; // groups whose strong masks took one launch / one launch per frame (rd_detector_counter 16 / 17)
  hipEvent_t last_strong; int have_last_strong;
  long next_enqueue, next_poll;
  long done_seq;                          // one more than the highest sequence number whose device work a worker has seen finished
  int last_polled_slot;
  void *last_segs; int last_nsegs;
  int use_graph, poly_mode, force_redo, fork_poly, fixed_rounds, budget_cycle; long n_redo, n_redo_rounds, n_redo_absorb;
  int overflow_streak;
  int poly_overflows;                     // set once two frames in a row overflowed the single-launch polyline kernel: later frames go multi-launch
  int rounds_budget, need_hist[64]; unsigned need_pos; long budget_count[RD_NBUDGETS], need_count[21];
  int graph_fork;            // the forked segment (one or two frames in flight: polyline chain beside the blur / region chain) as a captured graph too
  hipStream_t st_upload;     // group mode, host frames: the stream the frames travel on (created on first use, from the pool of high-priority streams)
  int post_helpers;          // helper threads armed by every poll (rd_post.c), 0 = none
  long host_enqueue_ns;      // wall time the caller spent inside rd_detector_enqueue
  long dev_us, dev_frames;   // sum over polled frames of (last kernel end - first kernel start) on the frame's stream, HIP events
  double tan_aov; int have_tan;    // what the workers use ahead of the poll that asks for the result
  pthread_mutex_t tan_mu; pthread_cond_t tan_cv;
  // Slots share streams (slot i uses the streams of slot i mod 4) and a worker thread may repeat part of its slot's frame on such a
  // stream while the enqueueing thread captures another slot's launch sequence on it: a capture would record the foreign launches, and
  // a stream that is capturing must not be synchronised.  launch_mu serialises captures against the launches of a repeat; repeats wait
  // on an event of their own (ev_redo), never on the stream.
  pthread_mutex_t launch_mu;
  const void *pinned_lo, *pinned_hi;        // the last caller buffer that was verified to be pinned host memory (RD_FRAME_HOST_PINNED)
  const void *probed[2]; int probed_pinned[2];      // RD_FRAME_HOST, one or two frames in flight: the last two frame pointers asked about (a loop alternates between its two pages) and the answer
  long n_frames_pinned, n_frames_copied;     // host frames that travelled straight from the caller's pinned memory / through the detector's own staging pages
  long n_unsettled;          // frames whose region merge was still changing after RD_REGION_MAX_LAUNCHES launches (none on any fixture)
  long n_truncated;          // frames with more segment records than the slots' probe buffers hold (maxrec_dev): probed again into a larger buffer
};

// The device planes of a slot.  With frames batched per launch (rd_detector::zb > 1) every slot's planes are carved out of one allocation at
// the same offsets, so that the planes of slot k + z lie a constant number of bytes (the slot pitch) behind those of slot k: a kernel of a
// group launch reaches frame z's planes as `pointer + blockIdx.z * pitch`.  Otherwise every plane is an allocation of its own.
struct PlaneAlloc {
  char *base; size_t at; int mode;       // mode 0: hipMalloc per plane; 1: carve from base; 2: count bytes only
  template <typename T> T *get(size_t n) {
    if (mode == 0) return dnew<T>(n);
    const size_t bytes = (((n ? n : 1) * sizeof(T)) + 255) & ~(size_t)255;
    T *p = mode == 1 ? (T *)(base + at) : (T *)nullptr;
    at += bytes;
    return p;
  }
  bool real() const { return mode != 2; }
};

static void slot_planes(rd_detector *d, Slot *s, PlaneAlloc &A) {
  const size_t N = (size_t)d->N;
  s->bgr = A.get<uint8_t>(N * 4);
  s->plab0 = A.get<uint32_t>(N); s->plab1 = A.get<uint32_t>(N); s->smooth = A.get<uint32_t>(N); s->quant = A.get<uint32_t>(N);
  for (int k = 0; k < 3; k++) { s->tr[k] = A.get<float>(N); s->fw[k] = A.get<float>(N); s->bw[k] = A.get<float>(N); s->hz[k] = A.get<float>(N); s->bl[k] = A.get<float>(N); }
  s->vxy = A.get<float>(N * 2); s->strength = A.get<float>(N); s->nms = A.get<float>(N);
  int **ip[] = { &s->i0, &s->i1, &s->mask0, &s->tidy, &s->label1, &s->strsum, &s->region, &s->rsize,
                 &s->boundarysrc, &s->boundary, &s->lsid, &s->region0 };
  for (size_t i = 0; i < sizeof(ip) / sizeof(ip[0]); i++) *ip[i] = A.get<int>(N);
  s->scratch2 = A.get<int>(N * 3 + 256);      // region_merge: the initial forest, flags + allow bytes, the second label plane of the rounds
  s->d2s = A.get<int>(RD_D2_SCRATCH_INTS(N));
  s->table = A.get<int>(N * 4); s->claim = A.get<int>(N); s->tlist = A.get<int>(N);
  if (A.real()) rdk::reduce_ls_init(s->st, s->table, s->claim, s->tlist, (int)(N * 4 / 5));
  s->e8 = A.get<int8_t>(N);
  s->strongbits = A.get<unsigned long long>((size_t)((d->iw + 63) / 64) * d->ih + 8);
  s->mmbits = A.get<unsigned long long>((size_t)((d->iw + 63) / 64) * d->ih + 8);
  s->ext = A.get<uint16_t>(N);
  { size_t a = rdk::iir_pass_scratch_floats(3, d->ih, d->iw), b = rdk::iir_pass_scratch_floats(3, d->iw, d->ih); s->tails = A.get<float>(a > b ? a : b); }
  s->flags = A.get<int>(16); if (A.real()) { RD_HIP(hipMemset(s->flags, 0, 16 * sizeof(int))); RD_HIP(hipStreamSynchronize(0)); }
  s->iir_chunked = 1;
  s->lslist = A.get<uint8_t>(N * 16);
  s->probes = A.get<int>((size_t)d->maxrec_dev * 15 * 6);
}

// Streams of the high-priority pool (see slot_alloc) are kept for the life of the process and handed to the next detector of the same device: a detector
// that is closed and another one opened (bench.py's side configurations after the headline) then runs on the same hardware queues instead of a second set.
static struct { pthread_mutex_t mu; hipStream_t st[64]; int dev[64]; int n; } stream_pool = { PTHREAD_MUTEX_INITIALIZER };
static hipStream_t pooled_stream(int device) {
  hipStream_t st = NULL;
  pthread_mutex_lock(&stream_pool.mu);
  for (int i = 0; i < stream_pool.n; i++) if (stream_pool.dev[i] == device) { st = stream_pool.st[i]; stream_pool.st[i] = stream_pool.st[stream_pool.n - 1]; stream_pool.dev[i] = stream_pool.dev[stream_pool.n - 1]; stream_pool.n--; break; }
  pthread_mutex_unlock(&stream_pool.mu);
  if (st) {      // (handed back idle - its last user synchronised it; one that reports an error instead is not handed out again)
    const hipError_t e = hipStreamQuery(st);
    if (e == hipSuccess || e == hipErrorNotReady) return st;
    (void)hipGetLastError();
    (void)hipStreamDestroy(st);
  }
  int lo = 0, hi = 0;
  RD_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
  RD_HIP(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi));
  return st;
}
static void unpool_stream(int device, hipStream_t st) {
  pthread_mutex_lock(&stream_pool.mu);
  const bool room = stream_pool.n < 64;
  if (room) { stream_pool.st[stream_pool.n] = st; stream_pool.dev[stream_pool.n] = device; stream_pool.n++; }
  pthread_mutex_unlock(&stream_pool.mu);
  if (!room) RD_HIP(hipStreamDestroy(st));
}

// the pool's teardown, for a caller that wants the streams gone (they are otherwise kept until the process ends, at most 64 of them): destroys every pooled stream
// of `device` (-1: of every device); streams in use by a live detector are not in the pool and are not touched
extern "C" void rd_release_cached_streams(int device) {
  hipStream_t gone[64]; int dev[64]; int n = 0;
  pthread_mutex_lock(&stream_pool.mu);
  for (int i = 0; i < stream_pool.n;) {
    if (device < 0 || stream_pool.dev[i] == device) {
      gone[n] = stream_pool.st[i]; dev[n] = stream_pool.dev[i]; n++;
      stream_pool.st[i] = stream_pool.st[stream_pool.n - 1]; stream_pool.dev[i] = stream_pool.dev[stream_pool.n - 1]; stream_pool.n--;
    } else i++;
  }
  pthread_mutex_unlock(&stream_pool.mu);
  for (int i = 0; i < n; i++) { RD_HIP(hipSetDevice(dev[i])); RD_HIP(hipStreamSynchronize(gone[i])); RD_HIP(hipStreamDestroy(gone[i])); }
}

// A slot's stream for repeats and fetches (created on first use): from the runtime's high-priority pool of hardware queues.  As ordinary streams they share the
// four queues of the default pool with the four streams that carry the groups, so a repeat - the oldest frame in flight, the one the caller waits for - stood in a queue
// behind whole groups of later frames after all (the reason it has a stream of its own); with queues of their own: 2853 against 2776-2791 frames/s on one box
// (RD_REDO_STREAM_PRIORITY=0: ordinary streams).  The group streams themselves stay in the default pool: moved to the high-priority pool they gain as much at 1920x1080
// but a detector opened after another one was closed in the same process (bench.py's side configurations) then ran 10-16 % slower - not understood, not kept.
static hipStream_t make_redo_stream() {
  static const int prio = RD_LAB_INT("RD_REDO_STREAM_PRIORITY", 1);
  hipStream_t st = NULL;
  if (prio) { int lo = 0, hi = 0; RD_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi)); RD_HIP(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi)); }
  else RD_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  return st;
}

// share: the slot whose streams this one uses as well (NULL: own streams)
static void slot_alloc(rd_detector *d, Slot *s, Slot *share) {
  const size_t N = (size_t)d->N;
  s->shares_streams = share != NULL;
  if (share) { s->st = share->st; s->st2 = share->st2; }
  else {
    // One or two frames in flight: each frame spreads over two streams, and the four streams of two frames need four hardware queues of their own.  The
    // runtime hands its four queues per priority level to the streams in the order they first launch something, and the null stream (set-up copies) and the
    // caller's command queue hold two of them: the second frame's streams then land on the queues of the first frame's - its main chain behind the other's polyline
    // chain (traced: two frames in flight ran 1.18 times as fast as one).  The streams of such a detector therefore come from the pool of another priority
    // level, where nothing else lives.  (Raising the number of queues for everybody - GPU_MAX_HW_QUEUES=8 - does the same for this path, 1420 -> 1590 frames/s, but
    // costs the group path, whose four streams are best served by four queues, 9 %.)
    static const int fork_prio = RD_LAB_INT("RD_FORK_STREAM_PRIORITY", 1);
    const bool from_pool = fork_prio != 0;
    if (d->fork_poly && from_pool) {
      s->st = pooled_stream(d->device);
      s->st2 = pooled_stream(d->device);
      s->pooled_streams = 1;
    } else {
    RD_HIP(hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking));
    RD_HIP(hipStreamCreateWithFlags(&s->st2, hipStreamNonBlocking));
    }
  }
  RD_HIP(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
  RD_HIP(hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
  RD_HIP(hipEventCreateWithFlags(&s->ev_mm, hipEventDisableTiming));
  RD_HIP(hipEventCreate(&s->ev_begin));
  RD_HIP(hipEventCreate(&s->ev_done));
  RD_HIP(hipEventCreateWithFlags(&s->ev_strong, hipEventDisableTiming));
  RD_HIP(hipEventCreateWithFlags(&s->ev_redo, hipEventDisableTiming));
  RD_HIP(hipEventCreateWithFlags(&s->ev_upload, hipEventDisableTiming));
  RD_HIP(hipEventCreateWithFlags(&s->ev_dense, hipEventDisableTiming));
  { PlaneAlloc A = { d->arena ? d->arena + (size_t)(s - d->slots) * d->slot_pitch : NULL, 0, d->arena ? 1 : 0 }; slot_planes(d, s, A); }
  s->ps = rdk::poly_scratch_create(d->iw, d->ih);
  RD_HIP(hipHostMalloc(&s->h_bgr, N * 4, hipHostMallocDefault));
  const size_t pack_ints = 64 + (size_t)RD_MAXREC * (14 + 15 * 6);
  RD_HIP(hipHostMalloc((void **)&s->h_pack, pack_ints * sizeof(int), hipHostMallocDefault)); memset(s->h_pack, 0, pack_ints * sizeof(int));
  RD_HIP(hipHostGetDevicePointer((void **)&s->h_pack_dev, s->h_pack, 0));
  s->h_ctr = s->h_pack; s->h_segs = s->h_pack + 64; s->h_probes = s->h_pack + 64 + (size_t)RD_MAXREC * 14;
  s->rounds = 20;
  s->seq = -1;
  if (d->device_post) {
    s->post_scratch = dnew<int>(rdk::post_scratch_ints());
    RD_HIP(hipHostMalloc((void **)&s->h_post, rdk::post_out_ints() * sizeof(int), hipHostMallocDefault)); memset(s->h_post, 0, rdk::post_out_ints() * sizeof(int));
    RD_HIP(hipHostGetDevicePointer((void **)&s->h_post_dev, s->h_post, 0));
  }
  // descriptor of the sparse stages
  s->frame = d->frames + (s - d->slots);
  rdk::PolyFrame &f = *s->frame;
  memset(&f, 0, sizeof(f));
  f.ps = *s->ps; f.in = NULL; f.in_bits = s->strongbits; f.ring_src = NULL; f.lslist = s->lslist; f.ids = s->lsid;
  f.boundary = s->boundary; f.table = s->table; f.claim = s->claim; f.tlist = s->tlist; f.probes = s->probes; f.pack = s->h_pack_dev; f.rflags = s->scratch2 + N;
  f.post_scratch = s->post_scratch; f.post_out = s->h_post_dev;
}

static void slot_free(Slot *s, int device) {
  RD_HIP(hipStreamSynchronize(s->st));
  void *all[] = { s->bgr, s->plab0, s->plab1, s->smooth, s->quant, s->vxy, s->strength, s->nms, s->i0, s->i1, s->mask0, s->tidy, s->label1, s->strsum,
                  s->region, s->rsize, s->scratch2, s->d2s, s->boundarysrc, s->boundary, s->lsid, s->table, s->claim, s->tlist, s->region0, s->probes, s->e8, s->strongbits, s->mmbits, s->ext, s->tails, s->flags, s->lslist };
  if (!s->owner->arena) {
  for (void *p : all) dfree(p);
  for (int k = 0; k < 3; k++) { dfree(s->tr[k]); dfree(s->fw[k]); dfree(s->bw[k]); dfree(s->hz[k]); dfree(s->bl[k]); }
  }
  rdk::poly_scratch_destroy(s->ps);
  RD_HIP(hipHostFree(s->h_bgr)); RD_HIP(hipHostFree(s->h_pack));
  dfree(s->post_scratch); if (s->h_post) RD_HIP(hipHostFree(s->h_post));
  RD_HIP(hipEventDestroy(s->ev_begin)); RD_HIP(hipEventDestroy(s->ev_done)); RD_HIP(hipEventDestroy(s->ev_strong));
  RD_HIP(hipEventDestroy(s->ev_fork)); RD_HIP(hipEventDestroy(s->ev_mm)); RD_HIP(hipEventDestroy(s->ev_join)); RD_HIP(hipEventDestroy(s->ev_redo)); RD_HIP(hipEventDestroy(s->ev_dense)); RD_HIP(hipEventDestroy(s->ev_upload));
  if (s->st_redo) RD_HIP(hipStreamDestroy(s->st_redo));
  dfree(s->big_probes);
  if (!s->shares_streams) {
    if (s->pooled_streams) { RD_HIP(hipStreamSynchronize(s->st2)); RD_HIP(hipStreamSynchronize(s->st)); unpool_stream(device, s->st2); unpool_stream(device, s->st); }
    else {
    RD_HIP(hipStreamDestroy(s->st2));
    RD_HIP(hipStreamDestroy(s->st));
    }
  }
}

// The device part of one frame (reference oclrect.c:235-381), enqueued on the slot's stream in three segments so that
// each can be captured into a hipGraph: [0] up to the first labelling, [1] up to the hand-over of the strong-edge mask
// to the next frame (quirk H1, needs an event wait before it and an event record after it), [2] the rest.
// last part of a frame: polylines of the strong edges, votes, probes, transfers.  mode 1 uses the single-launch polyline
// stage, which reports frames that do not fit its on-chip tables in counter 25; slot_postprocess() then repeats this
// part with mode 0.
static void frame_polyline(rd_detector *d, Slot *s, hipStream_t st, int mode) {
  // frame ring of the bridging step is "non-zero" on this path (oclrect.c:361, H3); the dense id plane is only produced on request (debug plane)
  rdk::polyline(st, s->frame, 1, d->N * 16, 1, 4.0f, 20, d->iw, d->ih, mode);
}

// segment / boundary votes (oclrect.c:365-367) and the probes the host needs (oclrect.c:1066-1098) for nb consecutive slots.
// The block the host needs - counters + round flags, the first RD_MAXREC segment records, their probes - is assembled by the
// sampling kernel directly in pinned host memory (0.2 MB of posted writes; frames with more records fetch the rest on demand):
// no copy launch at the end of the frame.
// tables_are_clean: frame_regions() ran just before (its last kernel undoes the previous entries of the vote tables)
// with_post: also the rectangles on the device, for the aperture the caller polled with last (the reference hands the aperture over with
// the poll, after the frame: a frame polled with another one, or the first frames of a stream, are post-processed on the host)
static void frames_votes(rd_detector *d, const rdk::PolyFrame *frames, int nb, hipStream_t st, int tables_are_clean, int with_post, double tan_aov = 0.0) {
  const int nentry = d->N * 4 / 5;
  rdk::reduce_ls(st, frames, nb, d->iw, d->ih, nentry, tables_are_clean);
  rdk::sample_segments(st, frames, nb, d->maxrec_dev, d->iw, d->ih, nentry, RD_MAXREC);
  if (with_post) rdk::post_device(st, frames, nb, d->maxrec_dev, d->iw, d->ih, tan_aov);      // (the caller's snapshot of the aperture: what the frames are labelled with)
}
// the aperture the device post-process of frames launched now runs with (set by polls and rd_detector_set_aperture, possibly on another thread)
static int aperture_snapshot(rd_detector *d, double *tan_out) {
  pthread_mutex_lock(&d->tan_mu);
  const int have = d->have_tan; *tan_out = d->tan_aov;
  pthread_mutex_unlock(&d->tan_mu);
  return have;
}
static void frame_votes(rd_detector *d, Slot *s, int tables_are_clean) { frames_votes(d, s->frame, 1, s->st, tables_are_clean, 0); }

static void frame_tail(rd_detector *d, Slot *s, int mode) {   // both, in order, on the slot's main stream (overflow redo)
  frame_polyline(d, s, s->st, mode);
  frame_votes(d, s, 0);
}

// regions, their sizes, absorption of small ones, boundaries and boundary components (oclrect.c:325-342).  Reads planes that
// nothing later in the frame modifies (quant, mergemask, label1, junction), so it can be repeated with a larger round budget.
static void frame_regions(rd_detector *d, Slot *s, hipStream_t st_over = NULL) {
  const int iw = d->iw, ih = d->ih, N = d->N;
  hipStream_t st = st_over ? st_over : s->st;
  // regions (oclrect.c:325-336)
  int *d2scratch = s->d2s;
  int marked = 0;
  rdk::region_merge(st, s->region0, s->scratch2, (const int *)s->quant, s->mmbits, s->strongbits, iw, ih, s->rounds,
                    s->rsize, &marked);   // H2: the sizes start from the junction counts (evaluated by the first kernel)
  rdk::region_size(st, s->rsize, s->region0, N, d2scratch + N, marked);      // (also strips the rounds' marks from the labels)
  rdk::despeckle2(st, s->region, s->region0, d2scratch, s->rsize, 16, iw, ih, 1, s->scratch2 + N + RD_REGION_STATUS_AT);   // (status words: they travel to the host with the round flags)

  // region boundaries and their components (oclrect.c:340-342)
  rdk::label8_boundary(st, s->boundary, s->boundarysrc, s->region, iw, ih, s->table, s->claim, s->tlist, 1, 0, RD_BOUNDARY_FLATTEN);   // (also undoes the previous frame's vote-table entries)
}

// votes and probes of a frame whose regions were computed again (a frame whose rectangles the device computes gets them again as well,
// from the new regions, for the aperture known now)
static void redo_votes(rd_detector *d, Slot *s, hipStream_t st) {
  int with_post = 0;
  if (s->post_mode) {
    pthread_mutex_lock(&d->tan_mu);
    with_post = d->have_tan; s->post_tan = d->tan_aov;
    pthread_mutex_unlock(&d->tan_mu);
  }
  frames_votes(d, s->frame, 1, st, 1, with_post, s->post_tan);
  s->post_mode = with_post;
}

// The absorption of small regions again for a frame the two fast launches could not finish (h_ctr[52] != 0: more undecided pixels than
// the single-block tail holds, or dependency chains that wind through more rows than it sweeps - frames that consist of small
// regions), by rounds over work lists until nothing changes, then everything downstream of it.  Runs on a stream of the slot's own
// that is never captured (the loop synchronises), after the frame's ev_done: nothing else touches the slot's planes.
static void frame_absorb_slow(rd_detector *d, Slot *s) {
  if (!s->st_redo) s->st_redo = make_redo_stream();
  hipStream_t st = s->st_redo;
  rdk::despeckle2_slow(st, s->region, s->region0, s->d2s, s->rsize, 16, d->iw, d->ih);
  rdk::label8_boundary(st, s->boundary, s->boundarysrc, s->region, d->iw, d->ih, s->table, s->claim, s->tlist, 1, 0, RD_BOUNDARY_FLATTEN);
  redo_votes(d, s, st);
  RD_HIP(hipStreamSynchronize(st));
  s->h_ctr[52] = 0;
}

// The whole frame-to-frame dependency chain in ONE launch (not captured: its planes change from frame to frame): this frame's strong mask
// (what oclrect.c:307-313 derives from the strength sums) from the sums + the strong mask of the frame before (H1, oclrect.c:274-275: the
// reference starts the sums from that mask; here the mask's element is added where a sum is read).  The same pass yields the edge mask at
// 500 of oclrect.c:277-284 and filters the labels at 2500 (filtering once at 2500 equals filtering at 500 and then at 2500, and both
// masks come from the unfiltered labels).
static void frame_strong(rd_detector *d, Slot *s, hipStream_t st) {
  const size_t N = (size_t)d->N;
  const long t = s->seq;
  s->prev_in = d->prev_ring + (size_t)(t % d->nring) * N;
  int8_t *out = d->prev_ring + (size_t)((t + 1) % d->nring) * N;
  rdk::strength_masks(st, NULL, out, NULL, s->e8, s->label1, s->strsum, d->t_edge, d->t_strong, d->iw, d->ih, s->prev_in, s->strongbits);      // (the mask as bytes for the next frame, as bits for this frame's polylines; nobody reads it as ints)
}

// gradient direction, re-packed blurred Lab, strength, non-max suppression (oclrect.c:251-258) of nz frames: one tile kernel that keeps the three
// intermediate planes on the chip (rd_k_front.hip: k_grad_nms); frame sizes its tiles do not cover take the three operators' kernels.
// taps: the intermediate planes are written as well (debug planes "plab1", "vxy", "strength")
static bool front_is_fused(const rd_detector *d) { return rdk::grad_nms_fits(d->iw, d->ih) != 0; }
static void frames_grad_nms(rd_detector *d, Slot *s, hipStream_t st, int nz, size_t zs, int taps = 0) {
  const int iw = d->iw, ih = d->ih;
  if (front_is_fused(d)) {
    if (taps) rdk::grad_nms(st, s->nms, s->bl, iw, ih, nz, zs, s->plab1, s->vxy, s->strength);
    else rdk::grad_nms(st, s->nms, s->bl, iw, ih, nz, zs);
    return;
  }
  rdk::edgevec(st, s->vxy, s->bl[0], iw, ih, s->plab1, s->bl[1], s->bl[2], nz, zs);
  rdk::edge_plab(st, s->strength, s->plab1, iw, ih, nz, zs);
  rdk::thinthres(st, s->nms, s->strength, s->vxy, iw, ih, nz, zs);
}

static void frame_segment(rd_detector *d, Slot *s, int ws, int seg, hipStream_t st_over = NULL) {
  const int iw = d->iw, ih = d->ih, N = d->N;
  hipStream_t st = st_over ? st_over : s->st;
  if (seg == 0) {

  // (colour conversion, oclrect.c:245: launched by enqueue_frame in front of this segment, outside the recorded graph - its source is the
  //  caller's device buffer when the frame is resident in HBM: no copy of the frame)
  // sigma=1 blur of L, a, b -> packed blurred Lab (oclrect.c:246-251)
  { const float *c[3] = { s->tr[0], s->tr[1], s->tr[2] }; rdk::iir_blur_pass(st, s->hz, c, s->fw, s->bw, 3, ih, iw, 1, s->tails, s->flags, 1); }    // along x (source: 16-bit fields)
  { const float *c[3] = { s->hz[0], s->hz[1], s->hz[2] }; rdk::iir_blur_pass(st, s->bl, c, s->fw, s->bw, 3, iw, ih, 0, s->tails, s->flags + 1); }   // along y
  // gradient direction (+ the packing of the blurred Lab, oclrect.c:251, on the way), strength, non-max suppression (oclrect.c:253-258)
  frames_grad_nms(d, s, st, 1, 0);

  // mask of positive responses and the rect-path tidy (oclrect.c:262-272)
  // components (background included) of the tidied mask; the tidy itself runs inside the labelling's tile kernel (and clears the
  // strength sums for the H1 segment); the walk to the roots happens in the first kernel of the next segment
  rdk::label8_tidy(st, s->label1, NULL, s->tidy, s->nms, s->strsum, iw, ih, 1);      // (the mask of positive responses, oclrect.c:262-264, is not stored: nothing reads it - debug plane "mask0" derives it from the responses)
  // strength sums per component (oclrect.c:274-275) - without the strong mask of the frame before (H1), which frame_strong() adds
  rdk::calc_strength(st, s->strsum, s->nms, s->label1, iw, ih, NULL, 1);
  return;
  }
  // Three chains leave this point and meet again before the region stage / the votes:
  //   main stream: edge-stopped blur x20 -> quantise -> despeckle                              (oclrect.c:286-303)
  //   2nd stream : junction counts of the filtered labels -> merge mask                      (oclrect.c:315-321)
  //                then the polyline stage, which needs nothing but the strong mask          (oclrect.c:361)
  // Inside a captured graph the streams become parallel branches.
  static const int fork_order = RD_LAB_INT("RD_FORK_ORDER", 1);      // (0: polyline chain launched first, 1: after the region stage, 2: before it)
  if (d->fork_poly) {      // (else: everything on the main stream, blur chain first)
    RD_HIP(hipEventRecord(s->ev_fork, st));
    RD_HIP(hipStreamWaitEvent(s->st2, s->ev_fork, 0));
    rdk::junction_bits(s->st2, (unsigned long long *)s->scratch2, s->strongbits, iw, ih);
    rdk::merge_mask(s->st2, s->mmbits, (const unsigned long long *)s->scratch2, iw, ih);
    RD_HIP(hipEventRecord(s->ev_mm, s->st2));
    if (fork_order == 0) {
    frame_polyline(d, s, s->st2, s->poly_mode);
    RD_HIP(hipEventRecord(s->ev_join, s->st2));
    }
  }

  // edge-preserving smoothing x10, quantise, despeckle (oclrect.c:286-303)
  rdk::blblur_extents(st, s->ext, s->e8, iw, ih);
  // (Two pairs per launch - halo of 8 cells, four passes through two LDS planes, with and without running sums - halve the launches and
  //  the HBM traffic of this stage and were measured 6 % SLOWER at full rate: 100 KB of LDS leave one 1024-thread block per CU and the
  //  extra barriers cost more than the saved traffic; the stage is bound by vector instructions, not by memory.  DESIGN.md.)
  { const uint32_t *src = s->plab0;     // ping-pong between i0 and smooth; the 10th pair lands in smooth
    for (int i = 0; i < 10; i++) { uint32_t *dst = (i & 1) ? s->smooth : (uint32_t *)s->i0; rdk::blblur_pair(st, dst, s->ext, src, iw, ih); src = dst; } }
  rdk::despeckle(st, s->quant, s->smooth, s->nms, iw, ih, 1);      // quantisation to 24 levels per field (oclrect.c:298) happens on the fly

  if (d->fork_poly) RD_HIP(hipStreamWaitEvent(st, s->ev_mm, 0));
  else {
    rdk::junction_bits(st, (unsigned long long *)s->scratch2, s->strongbits, iw, ih);
    rdk::merge_mask(st, s->mmbits, (const unsigned long long *)s->scratch2, iw, ih);
  }

  if (d->fork_poly && fork_order == 2) { frame_polyline(d, s, s->st2, s->poly_mode); RD_HIP(hipEventRecord(s->ev_join, s->st2)); }
  frame_regions(d, s);
  if (d->fork_poly && fork_order == 1) { frame_polyline(d, s, s->st2, s->poly_mode); RD_HIP(hipEventRecord(s->ev_join, s->st2)); }

  if (d->batch > 1) return;      // the sparse stages of the group's frames follow in one set of launches (sparse_launch)
  if (d->fork_poly) RD_HIP(hipStreamWaitEvent(st, s->ev_join, 0));
  else frame_polyline(d, s, st, s->poly_mode);
  frame_votes(d, s, 1);
}

// region-merge round budgets a frame can be launched with; the rounds stop changing anything after ~10 on typical frames
// and every launched round costs two dispatches even when it exits at once, so the budget follows what recent frames
// needed (+ margin).  A frame whose last launched round still changed something is repeated with the full budget
// (slot_postprocess), so the result never depends on the budget.
// how the polyline stage of the next frame is launched: the single-block kernel (1) until two frames in a row overflowed its on-chip tables
// (e.g. 3840x2160), from then on the ~85-launch form (0), which is also what repeats a frame the single-block kernel gave up on.  (One
// cooperative launch of several blocks per frame was built in round 3 and measured slower than the 85 launches: profiles/NOTES_r03.md.)
static int current_poly_mode(const rd_detector *d) {
  if (d->poly_mode == 0) return 0;
  return __atomic_load_n(&d->poly_overflows, __ATOMIC_RELAXED) ? 0 : 1;
}

static const int kRoundBudgets[RD_NBUDGETS] = { 8, 10, 12, 14, 16, 18, 20 };

static void run_segment(rd_detector *d, Slot *s, int ws, int seg, hipStream_t st_over = NULL) {      // st_over (segment 1 only): another stream than the slot's
  hipStream_t lst = st_over ? st_over : s->st;
  // (one or two frames in flight: kernel by kernel - the captured graph of the forked segment starts its second branch 170 us late, and the front segment, a straight
  //  line of ten kernels, is no faster as a graph either: 1615-1623 against 1636-1652 frames/s two deep; RD_GRAPH_FORK=1 brings both graphs back)
  if (!d->use_graph || (d->fork_poly && !d->graph_fork)) { frame_segment(d, s, ws, seg, lst); return; }
  hipGraphExec_t *ge = &s->gexec[seg];
  if (seg == 2) for (int k = 0; k < RD_NBUDGETS; k++) if (kRoundBudgets[k] == s->rounds) ge = &s->gexec2[k * 3 + (d->batch == 1 ? s->poly_mode : 0)];
  if (!*ge) {
    hipGraph_t g = NULL;
    pthread_mutex_lock(&d->launch_mu);
    RD_HIP(hipStreamBeginCapture(lst, hipStreamCaptureModeThreadLocal));
    frame_segment(d, s, ws, seg, lst);
    RD_HIP(hipStreamEndCapture(lst, &g));
    pthread_mutex_unlock(&d->launch_mu);
    RD_HIP(hipGraphInstantiate(ge, g, NULL, NULL, 0));
    RD_HIP(hipGraphDestroy(g));
  }
  RD_HIP(hipGraphLaunch(*ge, lst));
}

static void enqueue_frame(rd_detector *d, Slot *s, int ws) {
  if (d->use_graph && s->graph_ws != ws) {
    for (int k = 0; k < 3; k++) if (s->gexec[k]) { RD_HIP(hipGraphExecDestroy(s->gexec[k])); s->gexec[k] = NULL; }
    for (int k = 0; k < 3 * RD_NBUDGETS; k++) if (s->gexec2[k]) { RD_HIP(hipGraphExecDestroy(s->gexec2[k])); s->gexec2[k] = NULL; }
    s->graph_ws = ws;
  }
  RD_HIP(hipEventRecord(s->ev_begin, s->st));
  s->watch_begin = s->ev_begin; s->watch_done = s->ev_done;
  s->group_n = 1;
  rdk::bgr2plab_transposed(s->st, s->plab0, s->tr, s->src, d->iw, d->ih, ws);
  run_segment(d, s, ws, 0);
  if (d->have_last_strong) RD_HIP(hipStreamWaitEvent(s->st, d->last_strong, 0));
  frame_strong(d, s, s->st);
  RD_HIP(hipEventRecord(s->ev_strong, s->st));
  d->last_strong = s->ev_strong; d->have_last_strong = 1;
  s->rounds = d->fixed_rounds ? d->fixed_rounds : __atomic_load_n(&d->rounds_budget, __ATOMIC_RELAXED);
  if (d->budget_cycle) s->rounds = kRoundBudgets[2 + (int)((s->seq / d->budget_cycle) % (RD_NBUDGETS - 2))];      // (tests: a new graph every few frames)
  s->poly_mode = current_poly_mode(d);
  if (d->batch == 1) { double tn = 0; const int have = aperture_snapshot(d, &tn); s->post_mode = d->device_post && have; s->post_tan = tn; }
  for (int k = 0; k < RD_NBUDGETS; k++) if (kRoundBudgets[k] == s->rounds) d->budget_count[k]++;
  run_segment(d, s, ws, 2);
  if (d->batch > 1) { RD_HIP(hipEventRecord(s->ev_dense, s->st)); s->pending_sparse = 1; }
  else {
    if (s->post_mode) rdk::post_device(s->st, s->frame, 1, d->maxrec_dev, d->iw, d->ih, s->post_tan);      // (outside the captured graphs: the aperture is a launch argument)
    RD_HIP(hipEventRecord(s->ev_done, s->st));
  }
  rdrt::check_launch("rect frame");
}


// ---- group mode (rd_detector::zb > 1): the frames of zb consecutive slots in ONE set of launches, dense stages included (frame =
// blockIdx.z; the slots' planes lie slot_pitch bytes apart).  For frames so small that a launch does not fill the device (a 640x480
// plane is 75 tiles for 256 CUs): the frame rate is then set by the number of launches the four hardware queues get through, and a
// group needs as many as a single frame.  What cannot be shared is the short middle segment - a frame's strength sums start from
// the strong mask of the frame before it (H1) - which runs frame by frame between the two group segments, on the same stream.
static hipStream_t group_stream(rd_detector *d, int g0) { return d->slots[(g0 / d->zb) % (d->nstreams > 0 ? d->nstreams : 1)].st; }

static void group_segment(rd_detector *d, Slot *s, int nz, int seg, hipStream_t st) {      // s: the group's first slot
  const int iw = d->iw, ih = d->ih, N = d->N;
  const size_t zs = d->slot_pitch;
  if (seg == 0) {
    { const float *c[3] = { s->tr[0], s->tr[1], s->tr[2] }; rdk::iir_blur_pass(st, s->hz, c, s->fw, s->bw, 3, ih, iw, 1, s->tails, s->flags, 1, nz, zs); }
    { const float *c[3] = { s->hz[0], s->hz[1], s->hz[2] }; rdk::iir_blur_pass(st, s->bl, c, s->fw, s->bw, 3, iw, ih, 0, s->tails, s->flags + 1, 0, nz, zs); }
    frames_grad_nms(d, s, st, nz, zs);
    rdk::label8_tidy(st, s->label1, NULL, s->tidy, s->nms, s->strsum, iw, ih, 1, nz, zs);
    rdk::calc_strength(st, s->strsum, s->nms, s->label1, iw, ih, NULL, 1, nz, zs);
    return;
  }
  // seg 2: everything after the strong masks, in the order of a single frame on one stream (frame_segment without its fork)
  rdk::blblur_extents(st, s->ext, s->e8, iw, ih, nz, zs);
  { const uint32_t *src = s->plab0;
    for (int i = 0; i < 10; i++) { uint32_t *dst = (i & 1) ? s->smooth : (uint32_t *)s->i0; rdk::blblur_pair(st, dst, s->ext, src, iw, ih, nz, zs); src = dst; } }
  rdk::despeckle(st, s->quant, s->smooth, s->nms, iw, ih, 1, nz, zs);
  rdk::junction_bits(st, (unsigned long long *)s->scratch2, s->strongbits, iw, ih, nz, zs);
  rdk::merge_mask(st, s->mmbits, (const unsigned long long *)s->scratch2, iw, ih, nz, zs);
  int marked = 0;
  rdk::region_merge(st, s->region0, s->scratch2, (const int *)s->quant, s->mmbits, s->strongbits, iw, ih, s->rounds, s->rsize, &marked, nz, zs);
  rdk::region_size(st, s->rsize, s->region0, N, s->d2s + N, marked, nz, zs);
  rdk::despeckle2(st, s->region, s->region0, s->d2s, s->rsize, 16, iw, ih, 1, s->scratch2 + N + RD_REGION_STATUS_AT, nz, zs);
  rdk::label8_boundary(st, s->boundary, s->boundarysrc, s->region, iw, ih, s->table, s->claim, s->tlist, nz, zs, RD_BOUNDARY_FLATTEN);
  rdk::polyline(st, s->frame, nz, N * 16, 1, 4.0f, 20, iw, ih, s->poly_mode);
  frames_votes(d, s->frame, nz, st, 1, 0);
}

static void run_group_segment(rd_detector *d, Slot *s, int nz, int seg, hipStream_t st) {
  if (!d->use_graph) { group_segment(d, s, nz, seg, st); return; }
  hipGraphExec_t *ge = &s->gz0;
  if (seg == 2) for (int k = 0; k < RD_NBUDGETS; k++) if (kRoundBudgets[k] == s->rounds) ge = &s->gz2[k * 3 + s->poly_mode];
  if (!*ge) {
    hipGraph_t g = NULL;
    pthread_mutex_lock(&d->launch_mu);
    RD_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    group_segment(d, s, nz, seg, st);
    RD_HIP(hipStreamEndCapture(st, &g));
    pthread_mutex_unlock(&d->launch_mu);
    RD_HIP(hipGraphInstantiate(ge, g, NULL, NULL, 0));
    RD_HIP(hipGraphDestroy(g));
  }
  RD_HIP(hipGraphLaunch(*ge, st));
}

static void enqueue_frame(rd_detector *d, Slot *s, int ws);
static void slot_submitted(rd_detector *d, Slot *s);

// the frames waiting in the group that starts at slot g0: all zb of them with one row stride -> one set of launches; anything else (a poll
// that wants a result before the group is full, the end of a stream) -> frame by frame, the way a detector without groups launches them
static void group_launch(rd_detector *d, int g0) {
  const int zb = d->zb;
  int cnt = 0, same_ws = 1;
  for (int i = g0; i < g0 + zb && i < d->nslots; i++) { Slot *s = &d->slots[i]; if (s->pending_dense) { cnt++; if (s->ws != d->slots[g0].ws) same_ws = 0; } }
  if (cnt == 0) return;
  // Host frames travelled when they were handed over, one after the other on the detector's upload stream (rd_detector_enqueue).  The launching thread waits for that
  // stream here - at most for the transfer just issued, 0.1 ms, on a thread that has nothing else to do until the next group completes - and what the host has seen
  // complete needs no ordering on the device: no event.  Round 6 found the 7-8 % that host frames cost against frames resident in HBM in exactly two places: an event
  // recorded after every upload (a record is a system-scope release, a write-back of the device's caches, 2 900 times a second in the middle of everybody's kernels:
  // 2671-2698 frames/s with it, 2886-2901 without) and - pinned frames, whose hand-over takes microseconds - eight transfers queued on the copy engine at once
  // (2701-2723 against 2849-2884 with one at a time).  profiles/NOTES_r06.md.
  bool travelled = false;
  for (int i = g0; i < g0 + zb && i < d->nslots; i++) { Slot *s = &d->slots[i]; if (s->pending_dense && s->src == s->bgr && s->uploaded_early) travelled = true; }
  if (travelled) RD_HIP(hipStreamSynchronize(d->st_upload));
  if (cnt < zb || !same_ws) {
    for (int i = g0; i < g0 + zb && i < d->nslots; i++) {
      Slot *s = &d->slots[i];
      if (!s->pending_dense) continue;
      s->pending_dense = 0;
      if (s->src == s->bgr) { if (s->uploaded_early) { /* arrived: waited for above */ } else RD_HIP(hipMemcpyAsync(s->bgr, s->h_bgr, (size_t)s->ws * d->ih, hipMemcpyHostToDevice, s->st)); }
      enqueue_frame(d, s, s->ws);
      slot_submitted(d, s);
    }
    return;
  }
  Slot *lead = &d->slots[g0];
  hipStream_t st = group_stream(d, g0);
  const int ws = lead->ws;
  if (d->use_graph && lead->gz_ws != ws) {
    if (lead->gz0) { RD_HIP(hipGraphExecDestroy(lead->gz0)); lead->gz0 = NULL; }
    for (int k = 0; k < 3 * RD_NBUDGETS; k++) if (lead->gz2[k]) { RD_HIP(hipGraphExecDestroy(lead->gz2[k])); lead->gz2[k] = NULL; }
    lead->gz_ws = ws;
  }
  static const bool one_event = RD_LAB_INT("RD_GROUP_ONE_EVENT", 1) != 0;
  const uint8_t *srcs[RD_ZB_MAX];
  for (int i = 0; i < zb; i++) {
    Slot *s = &d->slots[g0 + i];
    s->pending_dense = 0;
    if (s->src == s->bgr && !s->uploaded_early) RD_HIP(hipMemcpyAsync(s->bgr, s->h_bgr, (size_t)ws * d->ih, hipMemcpyHostToDevice, st));
    srcs[i] = s->src;
    // (ONE pair of events brackets the group - its frames start and end together -, not one per frame: every record is a release fence in the stream)
    if (i == 0 || !one_event) RD_HIP(hipEventRecord(s->ev_begin, st));
    s->watch_begin = one_event ? lead->ev_begin : s->ev_begin; s->watch_done = one_event ? lead->ev_done : s->ev_done;
  }
  rdk::bgr2plab_transposed(st, lead->plab0, lead->tr, srcs, d->iw, d->ih, ws, zb, d->slot_pitch);
  run_group_segment(d, lead, zb, 0, st);
  // the strong masks: each frame's on top of its predecessor's (H1) - one launch for the group where its planes allow 16-byte accesses (the mask of the frame before
  // only decides sums that stand one below a threshold, and is then evaluated on the spot: k_strength_masks_group), else frame by frame
  if (d->have_last_strong) RD_HIP(hipStreamWaitEvent(st, d->last_strong, 0));      // (only the first frame waits for the group before - another stream - and only the last is waited for)
  bool consecutive = true;
  for (int i = 1; i < zb; i++) consecutive = consecutive && d->slots[g0 + i].seq == lead->seq + i;
  if (!d->strong_by_frame && consecutive && rdk::strength_masks_group_fits(d->iw, lead->label1, d->prev_ring, lead->e8, d->slot_pitch) && (d->N & 3) == 0) {
    for (int i = 0; i < zb; i++) d->slots[g0 + i].prev_in = d->prev_ring + (size_t)(d->slots[g0 + i].seq % d->nring) * (size_t)d->N;
    rdk::strength_masks_group(st, d->prev_ring, lead->e8, lead->label1, lead->strsum, d->t_edge, d->t_strong, d->iw, d->ih, lead->strongbits, lead->seq, d->nring, zb, d->slot_pitch);
    d->n_strong_group++;
  } else {
    for (int i = 0; i < zb; i++) frame_strong(d, &d->slots[g0 + i], st);
    d->n_strong_by_frame++;
  }
  { Slot *s = &d->slots[g0 + zb - 1]; RD_HIP(hipEventRecord(s->ev_strong, st)); d->last_strong = s->ev_strong; d->have_last_strong = 1; }
  const int rounds = d->fixed_rounds ? d->fixed_rounds : __atomic_load_n(&d->rounds_budget, __ATOMIC_RELAXED);
  const int pm = current_poly_mode(d);
  for (int i = 0; i < zb; i++) {
    Slot *s = &d->slots[g0 + i];
    s->rounds = rounds; s->poly_mode = pm; s->post_mode = 0;
    if (d->budget_cycle) s->rounds = kRoundBudgets[2 + (int)((lead->seq / d->budget_cycle) % (RD_NBUDGETS - 2))];
    for (int k = 0; k < RD_NBUDGETS; k++) if (kRoundBudgets[k] == s->rounds) d->budget_count[k]++;
  }
  run_group_segment(d, lead, zb, 2, st);
  double tn = 0;
  const int with_post = aperture_snapshot(d, &tn) && d->device_post;
  if (with_post) rdk::post_device(st, lead->frame, zb, d->maxrec_dev, d->iw, d->ih, tn);
  for (int i = 0; i < zb; i++) {
    Slot *s = &d->slots[g0 + i];
    s->post_mode = with_post; s->post_tan = tn; s->group_n = zb;
    if (i == 0 || !one_event) RD_HIP(hipEventRecord(s->ev_done, st));
  }
  rdrt::check_launch("rect frames, group launch");
  for (int i = 0; i < zb; i++) slot_submitted(d, &d->slots[g0 + i]);
}

// the frame is on its way: its worker thread may start waiting for ev_done (which has been recorded by now - an event that was never
// recorded counts as complete)
static void slot_submitted(rd_detector *d, Slot *s) {
  if (d->nworkers > 0) {
    pthread_mutex_lock(&s->mu);
    s->state = 1;
    pthread_cond_broadcast(&s->cv);
    pthread_mutex_unlock(&s->mu);
  }
}

// Batched mode: the sparse stages of slots a..b (consecutive slots of one group, dense stages enqueued) as one set of launches on
// the stream of one of them - a different one each time, so that the extra ~0.4 ms of queue time does not always delay the same
// stream's next frame - behind the dense stages of all of them.
static void sparse_launch(rd_detector *d, int a, int b) {
  const int nb = b - a + 1;
  Slot *host = &d->slots[a + (int)(d->sparse_rot++ % (unsigned)nb)];
  hipStream_t st = host->st;
  for (int i = a; i <= b; i++) if (d->slots[i].st != st) RD_HIP(hipStreamWaitEvent(st, d->slots[i].ev_dense, 0));
  const int pm = current_poly_mode(d);
  const rdk::PolyFrame *frames = d->frames + a;
  rdk::polyline(st, frames, nb, d->N * 16, 1, 4.0f, 20, d->iw, d->ih, pm);
  double tn = 0;
  const int with_post = aperture_snapshot(d, &tn) && d->device_post;
  frames_votes(d, frames, nb, st, 1, with_post, tn);
  for (int i = a; i <= b; i++) {
    Slot *s = &d->slots[i];
    s->post_mode = with_post; s->post_tan = tn;
    s->poly_mode = pm;
    RD_HIP(hipEventRecord(s->ev_done, st));
    s->watch_done = s->ev_done;
    s->pending_sparse = 0;
  }
  rdrt::check_launch("rect frames, sparse stages");
  for (int i = a; i <= b; i++) slot_submitted(d, &d->slots[i]);
}

// launches what is pending in the group of slot si (a poll needs one of its frames, or the detector is drained)
static void sparse_flush(rd_detector *d, int si) {
  const int g0 = si / d->batch * d->batch, g1 = g0 + d->batch - 1 < d->nslots - 1 ? g0 + d->batch - 1 : d->nslots - 1;
  int a = -1, b = -1;
  for (int i = g0; i <= g1; i++) if (d->slots[i].pending_sparse) { if (a < 0) a = i; b = i; }
  if (a >= 0) sparse_launch(d, a, b);
  if (d->deferred_slot >= g0 && d->deferred_slot <= g1) d->deferred_slot = -1;
}

static void wait_event_outside_captures(rd_detector *d, hipEvent_t ev);
static void slot_fetch(Slot *s, void *dst, const void *src, size_t bytes);

// What remains to be done on the device for a finished slot, once per frame (on the polling thread or on the slot's worker): the two
// rare repeats and the bookkeeping of the round budget.
static void slot_finish_device(rd_detector *d, Slot *s) {
  if (s->rounds <= 20 && s->h_ctr[32 + s->rounds - 1] != 0) {   // the region merge was still changing in its last launch: again with as many launches as it takes
    // (32 first - the frames that exceed a budget of 12-14 need 13-20 as a rule, and every launch after the settling one still costs a
    //  dispatch of the whole grid - then 64, then the definition's limit of RD_REGION_MAX_LAUNCHES = 128)
    // (on a stream of the slot's own: the slot's regular stream is one of the four that carry the groups, and the repeat would wait there behind a whole group of
    //  other frames; nothing but this frame's result depends on it - the frame is finished, ev_done has been waited for)
    static const bool redo_inline = RD_LAB_INT("RD_REDO_ON_MAIN_STREAM", 0) != 0;
    if (!s->st_redo && !redo_inline) s->st_redo = make_redo_stream();
    hipStream_t rst = redo_inline ? s->st : s->st_redo;
    for (int budget = 32; budget <= RD_REGION_MAX_LAUNCHES; budget *= 2) {
      s->rounds = budget;
      pthread_mutex_lock(&d->launch_mu);
      frame_regions(d, s, rst);
      redo_votes(d, s, rst);
      RD_HIP(hipEventRecord(s->ev_redo, rst));
      pthread_mutex_unlock(&d->launch_mu);
      wait_event_outside_captures(d, s->ev_redo);
      int still = 0;
      slot_fetch(s, &still, s->scratch2 + d->N + budget - 1, sizeof(int));      // the flag of the budget's last launch
      if (!still) break;
      if (budget == RD_REGION_MAX_LAUNCHES) __atomic_add_fetch(&d->n_unsettled, 1, __ATOMIC_RELAXED);      // (the definition's limit: the frame keeps what that many launches made of it - counter 6)
    }
    __atomic_add_fetch(&d->n_redo_rounds, 1, __ATOMIC_RELAXED);
  }
  if (s->h_ctr[52] != 0 || (d->force_redo & 2)) {   // the absorption's fast path gave up on this frame: finish it the long way
    frame_absorb_slow(d, s);
    __atomic_add_fetch(&d->n_redo_absorb, 1, __ATOMIC_RELAXED);
  }
  {   // budget for the frames to come (need = launches up to and including the first one that changed nothing)
    int need = 20;
    for (int r = 0; r < 20; r++) if (s->h_ctr[32 + r] == 0) { need = r + 1; break; }
    pthread_mutex_lock(&d->tan_mu);
    d->need_count[need]++;
    d->need_hist[d->need_pos++ & 63] = need;
    // what all but the three most demanding of the last 64 frames needed (a repeat costs a region stage, ~90 launches; a launch too many
    // costs 5 us in EVERY frame: on the bench stream 97 % of the frames need 9-11 launches, one in 150 needs 17 or more)
    int cnt[22] = { 0 }, seen = 0, mx = 20;
    for (int k = 0; k < 64; k++) cnt[d->need_hist[k] > 20 ? 20 : d->need_hist[k]]++;
    for (int v = 20; v >= 0; v--) { seen += cnt[v]; if (seen > 3) { mx = v; break; } }
    int b = 20;
    for (int k = RD_NBUDGETS - 1; k >= 0; k--) if (d->need_pos >= 8 && kRoundBudgets[k] >= mx + 1) b = kRoundBudgets[k];
    __atomic_store_n(&d->rounds_budget, b, __ATOMIC_RELAXED);
    pthread_mutex_unlock(&d->tan_mu);
  }
  // two overflows among the stream's recent frames (whichever slots they ran in): its frames do not fit the single-launch kernel (e.g. 4K) - stop trying
  if (s->poly_mode == 1 && s->h_ctr[25] != 0 && __atomic_add_fetch(&d->overflow_streak, 1, __ATOMIC_RELAXED) >= 2) __atomic_store_n(&d->poly_overflows, 1, __ATOMIC_RELAXED);
  if (s->poly_mode == 1 && s->h_ctr[25] == 0) __atomic_store_n(&d->overflow_streak, 0, __ATOMIC_RELAXED);
  if ((s->poly_mode && s->h_ctr[25] != 0) || (d->force_redo & 1)) {   // the single-launch polyline stage overflowed: repeat the tail the long way
    s->post_mode = 0;
    pthread_mutex_lock(&d->launch_mu);
    frame_tail(d, s, 0);
    RD_HIP(hipEventRecord(s->ev_redo, s->st));
    pthread_mutex_unlock(&d->launch_mu);
    wait_event_outside_captures(d, s->ev_redo);
    __atomic_add_fetch(&d->n_redo, 1, __ATOMIC_RELAXED);
  }
}

// Waiting for an event from a thread that is not the enqueueing one.  Slots share streams, and the enqueueing thread records a new graph
// on a slot's stream whenever a (slot, launch budget) pair is used for the first time - in the middle of a run, since the budget of the
// region merge follows the stream.  HIP refuses hipEventSynchronize / hipEventQuery on an event whose stream is being captured ("operation
// not permitted on an event last recorded in a capturing stream"), so the wait is a poll whose queries exclude captures (launch_mu: held by
// run_segment for the duration of a capture, a few hundred microseconds).
static void wait_event_outside_captures(rd_detector *d, hipEvent_t ev) {
  for (;;) {
    pthread_mutex_lock(&d->launch_mu);
    const hipError_t e = hipEventQuery(ev);
    pthread_mutex_unlock(&d->launch_mu);
    if (e == hipSuccess) return;
    if (e != hipErrorNotReady) { RD_HIP(e); }
    struct timespec ts = { 0, 30000 };
    nanosleep(&ts, NULL);
  }
}

// device -> host copy for a slot whose device work is complete, from any thread: on the slot's private stream (never the legacy stream -
// a plain hipMemcpy would make that depend on a stream another thread may be capturing a graph on)
static void slot_fetch(Slot *s, void *dst, const void *src, size_t bytes) {
  if (!s->st_redo) s->st_redo = make_redo_stream();
  RD_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s->st_redo));
  RD_HIP(hipStreamSynchronize(s->st_redo));
}

// host post-process of a slot whose device work is complete: rectangles for the given aperture + a copy of the segment list
// (can run again for another aperture: touches nothing on the device but, for frames with very many segments, two copies)
static void *slot_rectangles(rd_detector *d, Slot *s, double tanAOV, void **segs_out, int *nsegs_out) {
  int n = ((int *)s->h_segs)[0];
  // the rectangles the device computed (rd_k_post.hip), if this frame has them for this aperture and nothing overflowed there:
  // the valid candidates' records in candidate order (= the reference's list order)
  if (s->post_mode && s->post_tan == tanAOV && s->h_post[1] == 0 && n + 1 <= d->maxrec_dev) {
    const int nc = s->h_post[0];
    const char *recs = (const char *)(s->h_post + 8 + RD_POST_MAXC);
    int nv = 0;
    for (int c = 0; c < nc; c++) nv += s->h_post[8 + c] != 0;
    rect_t *ret = (rect_t *)calloc((size_t)nv + 1, sizeof(rect_t));
    int at = 1;
    for (int c = 0; c < nc; c++) if (s->h_post[8 + c] != 0) memcpy(&ret[at++], recs + (size_t)c * sizeof(rect_t), sizeof(rect_t));
    ret[0].nItems = nv + 1;
    // the segment list handed to rd_detector_last_segments: all n records, as on the host path - the pinned block holds the first
    // RD_MAXREC of them, a frame with more fetches the list from the device
    void *copy0 = malloc((size_t)(n + 1) * 56);
    if (n + 1 <= RD_MAXREC) memcpy(copy0, s->h_segs, (size_t)(n + 1) * 56);
    else slot_fetch(s, copy0, s->lslist, (size_t)(n + 1) * 56);
    *segs_out = copy0; *nsegs_out = n;
    __atomic_add_fetch(&d->n_post_device, 1, __ATOMIC_RELAXED);
    return ret;
  }
  __atomic_add_fetch(&d->n_post_host, 1, __ATOMIC_RELAXED);
  const void *segs = s->h_segs; const int *probes = s->h_probes;
  void *big_segs = NULL; int *big_probes = NULL;
  int maxrec = d->maxrec_dev < RD_MAXREC ? d->maxrec_dev : RD_MAXREC;
  if (n + 1 > maxrec) {   // rare: more segments than the fixed-size transfer covers
    const int *dev_probes = s->probes;
    if (n + 1 > d->maxrec_dev) {
      // more records than the slot's probe buffer holds (the list itself has the reference's capacity, 16N / 56 records, pl:456): the
      // probes of all of them are taken again into a buffer that grows on demand - nothing is ever dropped
      if (!s->st_redo) s->st_redo = make_redo_stream();
      if (s->big_probes_cap < n + 1) { dfree(s->big_probes); s->big_probes = dnew<int>((size_t)(n + 1) * 15 * 6); s->big_probes_cap = n + 1; }
      rdk::PolyFrame f = *s->frame;
      f.probes = s->big_probes; f.pack = NULL;
      rdk::sample_segments(s->st_redo, &f, 1, n + 1, d->iw, d->ih, d->N * 4 / 5, 0);
      rdrt::check_launch("probes of a frame with very many segments");
      dev_probes = s->big_probes;
      __atomic_add_fetch(&d->n_truncated, 1, __ATOMIC_RELAXED);      // (counter 10: frames that took this path)
    }
    big_segs = malloc((size_t)(n + 1) * 56); big_probes = (int *)malloc((size_t)(n + 1) * 15 * 6 * sizeof(int));
    slot_fetch(s, big_segs, s->lslist, (size_t)(n + 1) * 56);
    slot_fetch(s, big_probes, dev_probes, (size_t)(n + 1) * 15 * 6 * sizeof(int));
    segs = big_segs; probes = big_probes; maxrec = n + 1;
  }
  // RD_DIAG_NO_POST (diagnostics only): an empty rectangle list instead of the host post-process, to see whether a run is host-bound
  struct timespec tp0, tp1;
  clock_gettime(CLOCK_MONOTONIC, &tp0);
  void *r = rd_post_run(segs, maxrec, probes, d->iw, d->ih, tanAOV);
  clock_gettime(CLOCK_MONOTONIC, &tp1);
  __atomic_add_fetch(&d->host_post_ns, (tp1.tv_sec - tp0.tv_sec) * 1000000000L + (tp1.tv_nsec - tp0.tv_nsec), __ATOMIC_RELAXED);
  const int ns = n < maxrec ? n : maxrec - 1;
  void *copy = malloc((size_t)(ns + 1) * 56);
  memcpy(copy, segs, (size_t)(ns + 1) * 56);
  free(big_segs); free(big_probes);
  *segs_out = copy; *nsegs_out = ns;
  return r;
}

static void *slot_worker(void *arg) {
  Slot *s = (Slot *)arg;
  rd_detector *d = s->owner;
  RD_HIP(hipSetDevice(d->device));
  for (;;) {
    pthread_mutex_lock(&s->mu);
    while (s->state != 1 && !s->quit) pthread_cond_wait(&s->cv, &s->mu);
    if (s->quit) { pthread_mutex_unlock(&s->mu); return NULL; }
    pthread_mutex_unlock(&s->mu);
    // the aperture arrives with the poll (reference API); workers run ahead with the last one seen
    pthread_mutex_lock(&d->tan_mu);
    while (!d->have_tan && !s->quit) pthread_cond_wait(&d->tan_cv, &d->tan_mu);
    const double tan = d->tan_aov;
    pthread_mutex_unlock(&d->tan_mu);
    if (s->quit) return NULL;
    // Frames finish roughly in sequence order, and the caller collects them in that order: only the workers of the frames next in line watch their event
    // closely (a query every 30 us); the others - with 64 frames in flight, most of them, for most of the 25 ms their frame spends on the device - sleep in
    // longer steps until the front of finished frames comes within two groups of theirs.  (All of them polling was two million event queries a second
    // through the runtime's locks, next to the thread that launches.)
    {
      const long window = 2L * (d->zb > 1 ? d->zb : 4);
      while (!s->quit && s->seq >= __atomic_load_n(&d->done_seq, __ATOMIC_RELAXED) + window) { struct timespec ts = { 0, 250000 }; nanosleep(&ts, NULL); }
    }
    wait_event_outside_captures(d, s->watch_done);
    { long cur = __atomic_load_n(&d->done_seq, __ATOMIC_RELAXED); while (cur < s->seq + 1 && !__atomic_compare_exchange_n(&d->done_seq, &cur, s->seq + 1, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { } }
    void *segs = NULL; int ns = 0;
    slot_finish_device(d, s);
    void *r = slot_rectangles(d, s, tan, &segs, &ns);
    pthread_mutex_lock(&s->mu);
    s->result = r; s->result_tan = tan; s->res_segs = segs; s->res_nsegs = ns;
    s->state = 2;
    pthread_cond_broadcast(&s->cv);
    pthread_mutex_unlock(&s->mu);
  }
}

extern "C" {

rd_detector *rd_detector_create(int device, int iw, int ih, int nslots, int nworkers) {
  if (rd_device_count() <= 0) exitf(-1, "rd_detector_create: no HIP device available - this library has no CPU path\n");
  if (iw < 16 || ih < 16) exitf(-1, "rd_detector_create: frame %dx%d too small\n", iw, ih);
  if ((long long)iw * ih >= (1ll << 25)) exitf(-1, "rd_detector_create: frame %dx%d too large (pixel indices are kept below 2^25: hash keys and marked label words rely on the spare bits)\n", iw, ih);
  if (nslots < 1) nslots = 1;
  RD_HIP(hipSetDevice(device));
  rd_detector *d = (rd_detector *)calloc(1, sizeof(*d));
  d->magic = MAGIC_RECT; d->device = device; d->iw = iw; d->ih = ih; d->N = iw * ih; d->nslots = nslots; d->nworkers = nworkers;
  d->maxrec_dev = d->N * 16 / 56;
  if (d->maxrec_dev > 65536) d->maxrec_dev = 65536;      // the slots' probe buffers; frames with more records are probed again into a buffer that grows (slot_rectangles)
  { const int m = rd_env_int("RD_MAXREC_DEV", 0); if (m >= 16 && m < d->maxrec_dev) d->maxrec_dev = m; }      // (tests: exercise that path)
  d->nring = nslots + 1;
  d->prev_ring = dnew<int8_t>((size_t)d->N * d->nring);
  RD_HIP(hipMemset(d->prev_ring, 0, (size_t)d->N * d->nring));
  d->use_graph = rd_env("RD_NO_GRAPH") ? 0 : 1;
  d->graph_fork = RD_LAB_INT("RD_GRAPH_FORK", 0);
  d->poly_mode = rd_env("RD_POLY_MULTILAUNCH") ? 0 : 1;      // (tests: the ~85-launch form for every frame)
  d->force_redo = (rd_env("RD_POLY_FORCE_REDO") ? 1 : 0) | (rd_env("RD_ABSORB_FORCE_SLOW") ? 2 : 0);   // tests: every frame also takes the polyline / absorption fallback
  // candidate funnel + pose estimation on the device (rd_k_post.hip) instead of on one worker thread per frame slot: RD_DEVICE_POST=0|1 decides;
  // otherwise the host path - 0.3 ms of CPU time per 1080p frame, i.e. 0.6 of a core at 2000 frames/s: measured 2050 frames/s on 8 cores
  // as on 256, against 1830 for the device path - unless this process may run on one or two cores only
  if (rd_env("RD_DEVICE_POST")) d->device_post = rd_env_int("RD_DEVICE_POST", 0) != 0;
  else {
    cpu_set_t set;
    const int ncpu = sched_getaffinity(0, sizeof(set), &set) == 0 ? CPU_COUNT(&set) : 0;
    d->device_post = (nworkers > 0 && ncpu > 0 && ncpu <= 2) ? 1 : 0;
  }
  // The device runs four hardware queues side by side (more are time-sliced: measured 2x slower per frame).  With one or two
  // frames in flight a frame spreads over two streams (polyline chain beside the blur chain: shortest latency); from three
  // frames on every frame keeps to one stream, so that four frames occupy the four queues (highest throughput).
  d->fork_poly = nslots <= 2 ? 1 : 0;
  if (RD_LAB_INT("RD_FORK_POLY", 1) == 0) d->fork_poly = 0;      // (measurements: one stream per frame with one or two in flight as well)
  // One or two frames in flight and no worker threads = the reference's call sequence (executeOnce, enqueueTask / pollTask): the caller's thread runs the
  // post-process at the end of every frame's latency; helper threads share its pose estimations (rd_post.c), RD_POST_HELPERS=n overrides their number
  d->post_helpers = 0;
  if (nworkers == 0 && nslots <= 2) {
    cpu_set_t set;
    const int ncpu = sched_getaffinity(0, sizeof(set), &set) == 0 ? CPU_COUNT(&set) : 0;
    d->post_helpers = rd_env_int("RD_POST_HELPERS", ncpu >= 16 ? 4 : (ncpu >= 8 ? 2 : (ncpu >= 4 ? 1 : 0)));      // (1 / 3 / 5 / 7 helpers: 1.19 / 1.14-1.20 / 1.11 / 1.12-1.20 ms per executeOnce against 1.28 without)
    if (d->post_helpers > 0) rd_post_helpers_configure(d->post_helpers);
  }
  // round budget of the region merge: what the last 64 frames needed + margin (8 / 12 / 16 / 20 launched rounds; the rounds after
  // the merge has settled are no-ops, but each still costs two launches of a thousand blocks), frames that turn out to need more
  // are repeated with all 20; RD_REGION_ROUNDS_FIXED=8|12|16|20 pins the budget (20: never repeat anything).
  d->fixed_rounds = 0;
  if (rd_env("RD_REGION_ROUNDS_FIXED")) { const int r = rd_env_int("RD_REGION_ROUNDS_FIXED", 20); d->fixed_rounds = (r >= 8 && r <= 20 && !(r & 1)) ? r : 20; }
  d->rounds_budget = 20;
  // The single-block polyline kernel holds 16 384 live chain pixels; a 1920x1080 frame of the synthetic streams has ~11 000, frames of 3 megapixels and
  // more are beyond it as a rule: their streams start on the multi-launch form instead of overflowing - and being repeated - until the two-overflow rule
  // below finds that out (3840x2160: 12 of the first 16 frames).  Smaller frames that overflow anyway are still caught by that rule.
  d->poly_overflows = (long)iw * ih > 3000000L ? 1 : 0;
  d->t_edge = 500; d->t_strong = 2500;      // oclrect.c:277-284, 307-313
  if (rd_env("RD_TEST_THRESHOLDS")) {      // tests only: other thresholds, so that many strength sums stand exactly one below one (where the mask of the frame before decides, H1)
    int a = 0, b = 0;
    if (sscanf(rd_env("RD_TEST_THRESHOLDS"), "%d,%d", &a, &b) == 2 && a > 0 && b >= a) { d->t_edge = a; d->t_strong = b; }
  }
  d->strong_by_frame = rd_env("RD_STRONG_BY_FRAME") ? 1 : 0;
  d->budget_cycle = rd_env_int("RD_BUDGET_CYCLE", 0);      // tests: the launch budget changes every so many frames (12, 14, .. 20, 12, ..)
  pthread_mutex_init(&d->tan_mu, NULL); pthread_cond_init(&d->tan_cv, NULL);
  pthread_mutex_init(&d->launch_mu, NULL);
  d->slots = (Slot *)calloc((size_t)nslots, sizeof(Slot));
  // frames per set of sparse-stage launches: 4 from twelve frames in flight on - the deferral by one group below needs a third group of
  // slots to keep the streams fed, and without it a batch is a barrier per group (measured 10 % slower than no batching) - else every
  // frame on its own; RD_BATCH=1..4 overrides (tests: batching with any slot count)
  d->batch = nslots >= 24 ? 8 : (nslots >= 12 ? 4 : 1);      // (these stages are latency-bound chains of gathers: eight frames take a launch as long as four)
  if (rd_env("RD_BATCH")) { const int b = rd_env_int("RD_BATCH", 1); d->batch = b < 1 ? 1 : (b > RD_MAXB ? RD_MAXB : b); }
  if (d->fork_poly || d->batch > nslots) d->batch = d->fork_poly ? 1 : nslots;
  d->frames = (rdk::PolyFrame *)calloc((size_t)nslots, sizeof(rdk::PolyFrame));
  d->deferred_slot = -1;
  d->defer = (d->batch > 1 && nslots >= 3 * d->batch) ? 1 : 0;       // needs a third group of slots to keep the streams fed meanwhile
  // One stream per frame: the frames beyond the fourth queue up behind earlier ones on the same four streams (slot i uses the
  // streams of slot i mod 4) - a stream of its own would be time-sliced onto the same four hardware queues and stall frames
  // that have nothing to do with each other, while a queued frame keeps its queue busy as soon as its predecessor is done
  // (the host's turn-around between "frame polled" and "next frame enqueued" otherwise idles a quarter of the device).
  const int nstreams = RD_LAB_INT("RD_NSTREAMS", 4) < 1 ? 1 : (RD_LAB_INT("RD_NSTREAMS", 4) > 8 ? 8 : RD_LAB_INT("RD_NSTREAMS", 4));      // (measurements only - 3 / 5 / 6 streams: 2776-2786 / 2746-2753 / 2827-2847 frames/s against 2881-2889 with four, one box)
  d->nstreams = nstreams < nslots ? nstreams : nslots;
  // Group mode: four frames per launch from twelve frame slots on (three groups: one being filled, two in flight), two from six on, eight from 32 on
  // (640x480: 9100 frames/s with 16 slots in groups of four, 11700 with 32 in groups of eight; 1080p: no difference);
  // RD_ZBATCH=k overrides (k frames per launch, 2..8; 0 / 1: off).
  d->zb = nslots >= 32 ? 8 : (nslots >= 12 ? 4 : (nslots >= 6 ? 2 : 1));      // (always at least three or four groups: one being filled, the others in flight on the four streams)
  if ((long long)iw * ih > 1920ll * 1088) d->zb = 1;      // (launches of larger frames fill the device on their own: 3840x2160 measured 1 % slower in groups)
  if (rd_env("RD_ZBATCH")) { const int z = rd_env_int("RD_ZBATCH", 1); d->zb = z < 1 ? 1 : (z > RD_ZB_MAX ? RD_ZB_MAX : z); }
  if (d->fork_poly || d->zb > nslots || nstreams <= 0) d->zb = 1;
  if (d->zb > 1) { d->batch = 1; d->defer = 0; }      // (a group's sparse stages follow its dense stages on the same stream)
  if (d->zb > 1) {
    Slot tmp; memset(&tmp, 0, sizeof(tmp));
    PlaneAlloc A = { NULL, 0, 2 };
    slot_planes(d, &tmp, A);
    d->slot_pitch = A.at;
    d->arena = dnew<char>((size_t)nslots * d->slot_pitch);
  }
  for (int i = 0; i < nslots; i++) {
    Slot *s = &d->slots[i];
    slot_alloc(d, s, (!d->fork_poly && nstreams > 0 && i >= nstreams) ? &d->slots[i % nstreams] : NULL);
    s->owner = d;
    pthread_mutex_init(&s->mu, NULL); pthread_cond_init(&s->cv, NULL);
    if (nworkers > 0 && pthread_create(&s->th, NULL, slot_worker, s) != 0) exitf(-1, "rd_detector_create: cannot start worker thread\n");
  }
  d->last_polled_slot = -1;
  rdk::quant_lut_init(d->slots[0].st);
  RD_HIP(hipDeviceSynchronize());
  return d;
}

void rd_detector_destroy(rd_detector *d) {
  if (!d || d->magic != MAGIC_RECT) exitf(-1, "rd_detector_destroy: bad handle\n");
  RD_HIP(hipSetDevice(d->device));
  RD_HIP(hipDeviceSynchronize());
  for (int i = d->nslots - 1; i >= 0; i--) {      // (slots that borrowed streams go before the slots that own them)
    Slot *s = &d->slots[i];
    if (d->nworkers > 0) {
      pthread_mutex_lock(&s->mu); s->quit = 1; pthread_cond_broadcast(&s->cv); pthread_mutex_unlock(&s->mu);
      pthread_mutex_lock(&d->tan_mu); pthread_cond_broadcast(&d->tan_cv); pthread_mutex_unlock(&d->tan_mu);
      pthread_join(s->th, NULL);
    }
    free(s->result); free(s->res_segs);
    for (int k = 0; k < 3; k++) if (s->gexec[k]) RD_HIP(hipGraphExecDestroy(s->gexec[k]));
    for (int k = 0; k < 3 * RD_NBUDGETS; k++) if (s->gexec2[k]) RD_HIP(hipGraphExecDestroy(s->gexec2[k]));
    if (s->gz0) RD_HIP(hipGraphExecDestroy(s->gz0));
    for (int k = 0; k < 3 * RD_NBUDGETS; k++) if (s->gz2[k]) RD_HIP(hipGraphExecDestroy(s->gz2[k]));
    slot_free(s, d->device);
  }
  if (d->st_upload) { RD_HIP(hipStreamSynchronize(d->st_upload)); unpool_stream(d->device, d->st_upload); }
  free(d->slots);
  free(d->frames);
  dfree(d->prev_ring);
  dfree(d->arena);
  free(d->last_segs);
  d->magic = 0;
  free(d);
}

// a frame's way to the device: pieces copied into pinned memory (by the caller and whichever helper threads are awake), uploaded in order as they complete
struct UploadJob { char *dst; const char *src; char *dev; size_t bytes, piece; hipStream_t st; bool nt; int n, uploaded; int ready[64]; };
static void upload_copy_piece(void *ctx, int i) {
  UploadJob *u = (UploadJob *)ctx;
  const size_t o = (size_t)i * u->piece, m = u->bytes - o < u->piece ? u->bytes - o : u->piece;
  if (u->nt) rd_copy_to_staging(u->dst + o, u->src + o, m); else memcpy(u->dst + o, u->src + o, m);
  __atomic_store_n(&u->ready[i], 1, __ATOMIC_RELEASE);
}
static void upload_progress(void *ctx) {      // (caller's thread only)
  UploadJob *u = (UploadJob *)ctx;
  int e = u->uploaded;
  while (e < u->n && __atomic_load_n(&u->ready[e], __ATOMIC_ACQUIRE)) e++;
  if (e == u->uploaded) return;
  const size_t o = (size_t)u->uploaded * u->piece, end = (size_t)e * u->piece < u->bytes ? (size_t)e * u->piece : u->bytes;
  RD_HIP(hipMemcpyAsync(u->dev + o, u->dst + o, end - o, hipMemcpyHostToDevice, u->st));
  u->uploaded = e;
}

// is this host buffer page-locked (allocatePinnedMemory of oclhelper.h, rd_host_alloc, hipHostMalloc, hipHostRegister)?  One question to the runtime per buffer, not per frame.
static bool host_buffer_is_pinned(rd_detector *d, const void *frame) {
  for (int k = 0; k < 2; k++) if (d->probed[k] == frame) return d->probed_pinned[k] != 0;
  hipPointerAttribute_t at;
  const bool pinned = hipPointerGetAttributes(&at, frame) == hipSuccess && at.type == hipMemoryTypeHost;
  if (!pinned) (void)hipGetLastError();
  d->probed[1] = d->probed[0]; d->probed_pinned[1] = d->probed_pinned[0];
  d->probed[0] = frame; d->probed_pinned[0] = pinned ? 1 : 0;
  return pinned;
}

long rd_detector_enqueue(rd_detector *d, const void *frame, int ws, int on_device) {
  if (!d || d->magic != MAGIC_RECT) exitf(-1, "rd_detector_enqueue: bad handle\n");
  if (d->next_enqueue - d->next_poll >= d->nslots) exitf(-1, "rd_detector_enqueue: %d frames already in flight (poll first)\n", d->nslots);
  if ((size_t)ws * d->ih > (size_t)d->N * 4) exitf(-1, "rd_detector_enqueue: row stride %d too large for a %dx%d frame\n", ws, d->iw, d->ih);
  RD_HIP(hipSetDevice(d->device));
  struct timespec ts0; clock_gettime(CLOCK_MONOTONIC, &ts0);
  Slot *s = &d->slots[d->next_enqueue % d->nslots];
  s->seq = d->next_enqueue; s->ws = ws;
  const size_t bytes = (size_t)ws * d->ih;
  Slot *wait_upload = NULL;
  if (on_device == RD_FRAME_DEVICE) s->src = (const uint8_t *)frame;      // read where it lies (the caller keeps it valid until the frame's poll returned)
  else if (on_device == RD_FRAME_HOST_PINNED) {
    // The caller's buffer is page-locked and stays as it is until the frame's poll: the copy engine takes it from there.  (The reference copies every frame into its own
    // pinned page first, oclrect.c:1256 - 6 MB per 1920x1080 frame by the caller's thread, 16 GB/s at full rate on the thread that also launches everything.)
    if (!(frame >= d->pinned_lo && (const char *)frame + bytes <= (const char *)d->pinned_hi)) {      // (one look per buffer, not per frame: a capture loop reuses its pages)
      hipPointerAttribute_t at;
      if (hipPointerGetAttributes(&at, frame) != hipSuccess || at.type != hipMemoryTypeHost) { (void)hipGetLastError(); exitf(-1, "rd_detector_enqueue: RD_FRAME_HOST_PINNED needs pinned host memory (rd_host_alloc, allocatePinnedMemory, hipHostMalloc, hipHostRegister); %p is not\n", frame); }
      d->pinned_lo = frame; d->pinned_hi = (const char *)frame + bytes;
    }
    // (The colour conversion reading the pinned buffer over PCIe itself, without the copy engine: 1985-2009 frames/s against 2656-2659 - the kernel then runs at the link's
    //  rate, a millisecond per group, with its blocks resident all the while.  profiles/NOTES_r06.md.)
    hipStream_t ust = s->st;
    s->uploaded_early = 0;
    if (d->zb > 1) { if (!d->st_upload) d->st_upload = pooled_stream(d->device); ust = d->st_upload; }      // (group mode: on the detector's upload stream, from the high-priority pool; an ordinary stream shares a hardware queue with a group's: 2786-2796 frames/s, the caller waiting 0.35 ms per frame)
    if (d->zb > 1) RD_HIP(hipStreamSynchronize(ust));      // (one transfer in the copy engine's queue at a time: the caller waits for the frame before - 0.1 ms where it used to copy for 0.16 - see group_launch)
    RD_HIP(hipMemcpyAsync(s->bgr, frame, bytes, hipMemcpyHostToDevice, ust));
    if (d->zb > 1) s->uploaded_early = 1;      // (no event here: group_launch records ONE behind the uploads of all its frames - see there)
    s->src = s->bgr;
    d->n_frames_pinned++;
  }
  else if (on_device != RD_FRAME_HOST) exitf(-1, "rd_detector_enqueue: on_device = %d (0: host memory, 1: device memory, 2: pinned host memory)\n", on_device);
  else if (d->zb == 1 && host_buffer_is_pinned(d, frame)) {
    // The reference's call shape (oclrect_enqueueTask / executeOnce) on a buffer that happens to be page-locked - allocatePinnedMemory of oclhelper.h hands such memory out
    // (oclhelper.c:837-851, poly.cpp:68-69): the copy engine reads it in place, nothing is copied by the caller's thread, and the frame's kernels are launched behind the
    // transfer at once.  The reference's contract - the caller may reuse the buffer as soon as the call returns (oclrect.c:1256 copies it) - is kept by returning only when
    // the engine has read it: waited for at the END of this call (enqueue_done), after the frame's ~45 launches, by which time it has long happened (6 MB in 0.12 ms).
    RD_HIP(hipMemcpyAsync(s->bgr, frame, bytes, hipMemcpyHostToDevice, s->st));
    RD_HIP(hipEventRecord(s->ev_upload, s->st));
    s->src = s->bgr;
    wait_upload = s;
    d->n_frames_pinned++;
  }
  else if (d->zb == 1) {
    d->n_frames_copied++;
    // a single frame: the copy into pinned memory and the upload in pieces, so that a piece travels while the next is being copied (6 MB at 1920x1080:
    // the copy alone takes a fifth of a millisecond of the caller's latency).  With helper threads (armed here: the call that hands a frame over is followed by
    // the poll that waits for one) the pieces are copied side by side and uploaded in order as they complete.
    static const bool nt_copy = RD_LAB_INT("RD_NT_COPY", 1) != 0;
    static const int npieces_env = RD_LAB_INT("RD_UPLOAD_PIECES", 0);
    static const bool par_copy = RD_LAB_INT("RD_PARALLEL_COPY", 1) != 0;
    const bool helpers = d->post_helpers > 0 && par_copy;
    if (helpers) rd_post_helpers_arm();
    int npieces = npieces_env > 0 ? npieces_env : (helpers ? 16 : 4);
    if (npieces > 64) npieces = 64;
    UploadJob u;
    u.dst = (char *)s->h_bgr; u.src = (const char *)frame; u.dev = (char *)s->bgr; u.bytes = bytes; u.st = s->st; u.nt = nt_copy; u.uploaded = 0;
    u.piece = ((bytes + npieces - 1) / npieces + 4095) & ~(size_t)4095;
    u.n = (int)((bytes + u.piece - 1) / u.piece);
    for (int i = 0; i < u.n; i++) u.ready[i] = 0;
    rd_helpers_run(upload_copy_piece, &u, u.n, upload_progress);
    upload_progress(&u);
    if (u.uploaded != u.n) exitf(-1, "rd_detector_enqueue: internal error (pieces of the frame left behind)\n");
    s->src = s->bgr;
  } else {      // (group mode: the group's frames are uploaded together when it is launched)
    d->n_frames_copied++;
    static const bool nt_copy_g = RD_LAB_INT("RD_NT_COPY", 1) != 0;
    if (nt_copy_g) rd_copy_to_staging(s->h_bgr, frame, bytes); else memcpy(s->h_bgr, frame, bytes);
    s->src = s->bgr;
    // The frame travels NOW, on a stream of the detector's own (high-priority pool: a hardware queue nobody computes on), not when its group is launched: eight uploads in
    // front of a group's kernels kept that group's stream - a quarter of the device's queues - waiting for the copy engine for 1.6 of its 10.8 ms.  The group's stream
    // waits for the event instead (group_launch).  RD_UPLOAD_EARLY=0: as before.
    static const bool upload_early = RD_LAB_INT("RD_UPLOAD_EARLY", 1) != 0;
    s->uploaded_early = 0;
    if (upload_early) {
      if (!d->st_upload) d->st_upload = pooled_stream(d->device);
      RD_HIP(hipMemcpyAsync(s->bgr, s->h_bgr, bytes, hipMemcpyHostToDevice, d->st_upload));
      s->uploaded_early = 1;      // (no event here: group_launch records ONE behind the uploads of all its frames)
    }
  }
  if (d->zb > 1) {      // group mode: launched together with the other frames of its group, once that is full (or a poll needs one of them)
    const int si = (int)(s - d->slots);
    s->pending_dense = 1;
    // (the last group of slots may be short - nslots need not be a multiple of zb - and is launched when ITS last slot is filled: every group
    //  is launched the moment its last frame arrives, so frames reach the device in sequence order and at most one group is ever waiting)
    if (si % d->zb == d->zb - 1 || si == d->nslots - 1) group_launch(d, si / d->zb * d->zb);
  } else {
  enqueue_frame(d, s, ws);
  }
  if (d->zb > 1) ;
  else if (d->batch == 1) slot_submitted(d, s);
  else {
    const int si = (int)(s - d->slots);
    if (si % d->batch == d->batch - 1 || si == d->nslots - 1) {      // the group is complete
      // Launched right away, the sparse stages would sit in their stream between this group's dense stages and the next one's, waiting
      // for the slowest of the group's four streams: a barrier per group (measured: 10 % slower than no batching).  One group later,
      // everything they wait for is long done and the stream they land on has the next group's dense work queued in front of them.
      if (d->defer) {
        const int prev = d->deferred_slot;
        d->deferred_slot = si;
        if (prev >= 0) sparse_flush(d, prev);
      } else sparse_flush(d, si);
    }
  }
  if (wait_upload) RD_HIP(hipEventSynchronize(wait_upload->ev_upload));      // (the caller may touch its buffer again)
  { struct timespec ts1; clock_gettime(CLOCK_MONOTONIC, &ts1); d->host_enqueue_ns += (ts1.tv_sec - ts0.tv_sec) * 1000000000L + (ts1.tv_nsec - ts0.tv_nsec); }
  return d->next_enqueue++;
}

void *rd_detector_poll(rd_detector *d, double tanAOV) {
  if (!d || d->magic != MAGIC_RECT) exitf(-1, "rd_detector_poll: bad handle\n");
  if (d->next_poll >= d->next_enqueue) exitf(-1, "rd_detector_poll: nothing enqueued\n");
  RD_HIP(hipSetDevice(d->device));
  const int si = (int)(d->next_poll % d->nslots);
  Slot *s = &d->slots[si];
  void *r = NULL, *segs = NULL; int ns = 0;
  if (s->pending_sparse) sparse_flush(d, si);       // (an incomplete group: the caller wants a result before handing over more frames)
  if (s->pending_dense) group_launch(d, si / d->zb * d->zb);
  pthread_mutex_lock(&d->tan_mu);
  d->tan_aov = tanAOV; d->have_tan = 1;            // what workers and the device post-process of later frames run ahead with
  pthread_cond_broadcast(&d->tan_cv);
  pthread_mutex_unlock(&d->tan_mu);
  if (d->nworkers > 0) {
    pthread_mutex_lock(&s->mu);
    while (s->state != 2) pthread_cond_wait(&s->cv, &s->mu);
    r = s->result; segs = s->res_segs; ns = s->res_nsegs;
    const double used = s->result_tan;
    s->result = NULL; s->res_segs = NULL; s->state = 0;
    pthread_mutex_unlock(&s->mu);
    if (used != tanAOV) {      // the worker ran ahead with another aperture: redo with the requested one
      free(r); free(segs);
      r = slot_rectangles(d, s, tanAOV, &segs, &ns);
    }
  } else {
    if (d->post_helpers && !(s->post_mode && s->post_tan == tanAOV)) rd_post_helpers_arm();      // (they wake while the device is still busy with the frame: rd_post.c; not for a frame whose rectangles the device computes)
    static const int trace_poll = RD_LAB_INT("RD_TRACE_POLL", 0);      // (1: timings, 2: wait by querying instead of hipEventSynchronize)
    struct timespec t0, t1, t2, t3; clock_gettime(CLOCK_MONOTONIC, &t0);
    if (trace_poll == 2) { while (hipEventQuery(s->watch_done) == hipErrorNotReady) sched_yield(); }
    else RD_HIP(hipEventSynchronize(s->watch_done));
    clock_gettime(CLOCK_MONOTONIC, &t1);
    slot_finish_device(d, s);
    clock_gettime(CLOCK_MONOTONIC, &t2);
    r = slot_rectangles(d, s, tanAOV, &segs, &ns);
    clock_gettime(CLOCK_MONOTONIC, &t3);
    if (trace_poll) {
      static double a, b, c; static int n;
      a += (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3; b += (t2.tv_sec - t1.tv_sec) * 1e6 + (t2.tv_nsec - t1.tv_nsec) * 1e-3; c += (t3.tv_sec - t2.tv_sec) * 1e6 + (t3.tv_nsec - t2.tv_nsec) * 1e-3;
      if (++n % 100 == 0) { fprintf(stderr, "poll: wait %.1f us, finish %.1f us, rectangles %.1f us (average of 100)\n", a / 100, b / 100, c / 100); a = b = c = 0; }
    }
  }
  { float ms = 0.0f; if (hipEventElapsedTime(&ms, s->watch_begin, s->watch_done) == hipSuccess) { d->dev_us += (long)(ms * 1000.0f) / (s->group_n > 0 ? s->group_n : 1); d->dev_frames++; } }      // (a group's interval is shared by its frames: counted once)
  free(d->last_segs);
  d->last_segs = segs; d->last_nsegs = ns;
  d->last_polled_slot = si;
  d->next_poll++;
  return r;
}

// The reference hands the aperture over with the poll, i.e. after the frame (oclrect_pollTask); whatever runs ahead of the poll - the
// worker threads' post-process, the rectangles on the device - uses the last one seen.  A caller that knows it beforehand says so here,
// and the first frames of a stream are treated like all later ones.
void rd_detector_set_aperture(rd_detector *d, double tanAOV) {
  if (!d || d->magic != MAGIC_RECT) exitf(-1, "rd_detector_set_aperture: bad handle\n");
  pthread_mutex_lock(&d->tan_mu);
  d->tan_aov = tanAOV; d->have_tan = 1;
  pthread_cond_broadcast(&d->tan_cv);
  pthread_mutex_unlock(&d->tan_mu);
}

void rd_detector_drain(rd_detector *d) {
  if (!d || d->magic != MAGIC_RECT) exitf(-1, "rd_detector_drain: bad handle\n");
  RD_HIP(hipSetDevice(d->device));
  if (d->batch > 1) for (int i = 0; i < d->nslots; i += d->batch) sparse_flush(d, i);
  if (d->zb > 1) for (long q = d->next_poll; q < d->next_enqueue; q++) { const int si = (int)(q % d->nslots); if (d->slots[si].pending_dense) group_launch(d, si / d->zb * d->zb); }      // (sequence order)
  for (int i = 0; i < d->nslots; i++) RD_HIP(hipStreamSynchronize(d->slots[i].st));
}

long rd_detector_counter(rd_detector *d, int which) {
  if (!d || d->magic != MAGIC_RECT) exitf(-1, "rd_detector_counter: bad handle\n");
  if (which == 3) return d->host_enqueue_ns / 1000;
  if (which == 4) return __atomic_load_n(&d->n_redo_rounds, __ATOMIC_RELAXED);
  if (which == 5) return __atomic_load_n(&d->rounds_budget, __ATOMIC_RELAXED);
  if (which >= 20 && which < 20 + RD_NBUDGETS) return d->budget_count[which - 20];   // frames launched with a budget of 8 / 10 / .. / 20 launches of the region merge
  if (which == 10) return __atomic_load_n(&d->n_truncated, __ATOMIC_RELAXED);
  if (which == 11) return __atomic_load_n(&d->n_post_device, __ATOMIC_RELAXED);
  if (which == 12) return __atomic_load_n(&d->n_post_host, __ATOMIC_RELAXED);
  if (which == 13) return __atomic_load_n(&d->host_post_ns, __ATOMIC_RELAXED) / 1000;
  if (which == 14) return __atomic_load_n(&d->n_redo_absorb, __ATOMIC_RELAXED);
  if (which == 15) return d->zb;      // frames per group launch (1: every frame its own launches)
  if (which == 6) return __atomic_load_n(&d->n_unsettled, __ATOMIC_RELAXED);
  if (which == 18) return d->n_frames_pinned;
  if (which == 19) return d->n_frames_copied;
  if (which == 16) return d->n_strong_group;        // groups whose strong masks were ONE launch (k_strength_masks_group)
  if (which == 17) return d->n_strong_by_frame;     // groups whose strong masks were evaluated frame by frame
  if (which >= 40 && which <= 60) return d->need_count[which - 40];      // frames whose region merge needed 0..20 launches (the one that changes nothing included; 20: or more)
  if (which == 1) return d->dev_us;
  if (which == 2) return d->dev_frames;
  return which == 0 ? __atomic_load_n(&d->n_redo, __ATOMIC_RELAXED) : -1;
}

int rd_detector_last_segments(rd_detector *d, void *dst, int max_records) {
  if (!d || d->magic != MAGIC_RECT) exitf(-1, "rd_detector_last_segments: bad handle\n");
  if (!d->last_segs) return -1;
  int m = d->last_nsegs + 1 < max_records ? d->last_nsegs + 1 : max_records;
  if (dst && m > 0) memcpy(dst, d->last_segs, (size_t)m * 56);
  return d->last_nsegs;
}

size_t rd_detector_debug_plane(rd_detector *d, const char *name, void *dst, size_t max_bytes) {
  if (!d || d->magic != MAGIC_RECT) exitf(-1, "rd_detector_debug_plane: bad handle\n");
  if (d->last_polled_slot < 0) return 0;
  RD_HIP(hipSetDevice(d->device));
  Slot *s = &d->slots[d->last_polled_slot];
  const size_t N = (size_t)d->N;
  struct { const char *n; const void *p; size_t bytes; } tab[] = {
    { "plab0", s->plab0, N * 4 }, { "plab1", s->plab1, N * 4 }, { "lblur", s->bl[0], N * 4 }, { "vxy", s->vxy, N * 8 }, { "strength", s->strength, N * 4 },
    { "nms", s->nms, N * 4 }, { "mask0", s->nms, N * 4 }, { "tidy", s->tidy, N * 4 }, { "label1", s->label1, N * 4 }, { "strsum", s->strsum, N * 4 },
    { "edge500", s->e8, N }, { "smooth", s->smooth, N * 4 }, { "quant", s->quant, N * 4 }, { "strong", s->strongbits, (size_t)((d->iw + 63) / 64) * d->ih * 8 }, { "junction", s->strongbits, (size_t)((d->iw + 63) / 64) * d->ih * 8 },
    { "mergemask", s->mmbits, (size_t)((d->iw + 63) / 64) * d->ih * 8 }, /* (bit planes: handed out as int planes by the branch below) */ { "region", s->region, N * 4 }, { "region0", s->region0, N * 4 }, { "rsize", s->rsize, N * 4 }, { "boundarysrc", s->boundarysrc, N * 4 },
    { "boundary", s->boundary, N * 4 }, { "lsid", s->lsid, N * 4 }, { "table", s->table, (N * 4 / 5) * 5 * 4 }, { "lslist", s->lslist, N * 16 }, { "polyctr", rdk::poly_scratch_counters(s->ps), 64 * 4 }, { "iirflags", s->flags, 16 * 4 }, { "d2work", s->d2s + N, 16 * 4 }, { "absorb", s->scratch2 + N + RD_REGION_STATUS_AT, 8 * 4 },
  };
  for (size_t i = 0; i < sizeof(tab) / sizeof(tab[0]); i++)
    if (!strcmp(tab[i].n, name)) {
      const size_t b = tab[i].bytes < max_bytes ? tab[i].bytes : max_bytes;
      // (launches of a test tap: under the lock that keeps launches out of another thread's graph capture)
      pthread_mutex_lock(&d->launch_mu);
      if (!strcmp(name, "lsid")) rdk::polyline_ids(s->st, s->frame, 1, (int)N);   // not part of the frame path: built from the compact state
      if (!strcmp(name, "boundary") && !RD_BOUNDARY_FLATTEN) rdk::label8_flatten(s->st, s->boundary, (int)N);   // the frame path leaves the components as a forest (its few readers walk): flattened here, where the plane is looked at
      if ((!strcmp(name, "plab1") || !strcmp(name, "vxy") || !strcmp(name, "strength")) && front_is_fused(d))
        frames_grad_nms(d, s, s->st, 1, 0, 1);      // these never leave the chip on the frame path: the same kernel again, writing them out (the blurred planes are intact)
      pthread_mutex_unlock(&d->launch_mu);
      RD_HIP(hipStreamSynchronize(s->st));
      if (!strcmp(name, "strong") || !strcmp(name, "mergemask") || !strcmp(name, "junction")) {
        // kept as bit planes on the device (what the polyline stage traces / what the region stage reads); handed out as the int planes of
        // oclrect.c:307-321 - the junction counts (oclrect.cl:74-95: on-pixels of the 3x3 block, 1 -> 0, frame border 0) evaluated here from the strong mask
        const size_t n = N * 4 <= max_bytes ? N : max_bytes / 4;
        const int wpr = (d->iw + 63) / 64, iw = d->iw, ih = d->ih;
        unsigned long long *tmp = (unsigned long long *)malloc((size_t)wpr * ih * 8 + 8);
        if (!tmp) exitf(-1, "rd_detector_debug_plane: out of memory\n");
        RD_HIP(hipMemcpy(tmp, tab[i].p, (size_t)wpr * ih * 8, hipMemcpyDeviceToHost));
        auto bit = [&](int x, int y) { return (int)((tmp[(size_t)y * wpr + (x >> 6)] >> (x & 63)) & 1ull); };
        for (size_t k = 0; k < n; k++) {
          const int y = (int)(k / iw), x = (int)(k % iw);
          int v = bit(x, y);
          if (!strcmp(name, "junction")) {
            if (v && x > 0 && y > 0 && x < iw - 1 && y < ih - 1) { int c = 0; for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) c += bit(x + dx, y + dy); v = c == 1 ? 0 : c; }
            else v = 0;
          }
          ((int *)dst)[k] = v;
        }
        free(tmp);
        return n * 4;
      }
      if (!strcmp(name, "mask0")) {         // not stored on the frame path: the mask of positive responses (oclrect.c:262-264), from the suppressed strength
        const size_t n = N * 4 <= max_bytes ? N : max_bytes / 4;
        float *tmp = (float *)malloc(n ? n * 4 : 4);
        if (!tmp) exitf(-1, "rd_detector_debug_plane: out of memory\n");
        RD_HIP(hipMemcpy(tmp, s->nms, n * 4, hipMemcpyDeviceToHost));
        for (size_t k = 0; k < n; k++) ((int *)dst)[k] = tmp[k] > 0.0f ? 1 : 0;
        free(tmp);
        return n * 4;
      }
      if (!strcmp(name, "edge500")) {       // kept as bytes on the device (the blur's mask); handed out as the int plane of oclrect.c:277-284
        const size_t n = N * 4 <= max_bytes ? N : max_bytes / 4;
        int8_t *tmp = (int8_t *)malloc(n ? n : 1);
        RD_HIP(hipMemcpy(tmp, tab[i].p, n, hipMemcpyDeviceToHost));
        for (size_t k = 0; k < n; k++) ((int *)dst)[k] = tmp[k];
        free(tmp);
        return n * 4;
      }
      RD_HIP(hipMemcpy(dst, tab[i].p, b, hipMemcpyDeviceToHost));
      if (!strcmp(name, "strsum") && s->prev_in) {      // the reference's plane holds the sums ON TOP of the previous frame's strong mask (H1): added here, where it is looked at
        const size_t n = b / 4;
        int8_t *tmp = (int8_t *)malloc(n ? n : 1);
        RD_HIP(hipMemcpy(tmp, s->prev_in, n, hipMemcpyDeviceToHost));
        for (size_t k = 0; k < n; k++) ((int *)dst)[k] += tmp[k];
        free(tmp);
      }
      return b;
    }
  return 0;
}

// ================================================================================================ oclrect (reference API)
struct oclrect_t { uint32_t magic; rd_detector *det; int iw, ih; };

struct oclrect_t *init_oclrect(struct oclimgutil_t *oclimgutil, struct oclpolyline_t *oclpolyline, cl_device_id device, cl_context context, cl_command_queue queue, int iw, int ih) {
  (void)oclimgutil; (void)oclpolyline; (void)context; (void)queue;
  struct oclrect_t *t = (struct oclrect_t *)calloc(1, sizeof(*t));
  t->magic = MAGIC_RECT; t->iw = iw; t->ih = ih;
  t->det = rd_detector_create(device ? device->ordinal : rdrt::current_device(), iw, ih, 2, RD_LAB_INT("RD_API_WORKERS", 0));   // two pages like oclrect.c:54
  return t;
}

void dispose_oclrect(struct oclrect_t *t) {
  if (!t || t->magic != MAGIC_RECT) exitf(-1, "dispose_oclrect: bad handle\n");
  rd_detector_destroy(t->det);
  t->magic = 0;
  free(t);
}

rect_t *oclrect_executeOnce(struct oclrect_t *t, uint8_t *imgData, int ws, const double tanAOV) {
  if (!t || t->magic != MAGIC_RECT) exitf(-1, "oclrect_executeOnce: bad handle\n");
  if (t->det->next_enqueue != t->det->next_poll) exitf(-1, "oclrect_executeOnce: a task is still pending (poll it first)\n");
  rd_detector_enqueue(t->det, imgData, ws, 0);
  return (rect_t *)rd_detector_poll(t->det, tanAOV);
}

void oclrect_enqueueTask(struct oclrect_t *t, uint8_t *imgData, int ws) {
  if (!t || t->magic != MAGIC_RECT) exitf(-1, "oclrect_enqueueTask: bad handle\n");
  rd_detector_enqueue(t->det, imgData, ws, 0);
}

rect_t *oclrect_pollTask(struct oclrect_t *t, const double tanAOV) {
  if (!t || t->magic != MAGIC_RECT) exitf(-1, "oclrect_pollTask: bad handle\n");
  return (rect_t *)rd_detector_poll(t->det, tanAOV);
}

}  // extern "C"
