/*
 * TEST INFRASTRUCTURE ONLY (oracle/). Never linked into the product library.
 *
 * OpenCV-free harness around the UNCHANGED reference host C.  The reference's demo programs
 * need OpenCV (absent here), so their call sequences are restated on raw BGR buffers:
 *   rdref_poly_run      <- poly.cpp:68-131   (buffer set-up, op sequence, read-back)
 *   rdref_rect_*        <- rect.cpp:78-105 / vidrect.cpp:128-172 (init, executeOnce, enqueue/poll)
 * Everything these functions call (oclimgutil_*, oclpolyline_execute, oclrect_*) is the
 * reference's own code compiled from /root/reference by oracle/Makefile.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CL_TARGET_OPENCL_VERSION 120
#define CL_USE_DEPRECATED_OPENCL_1_2_APIS
#include <CL/cl.h>

#include "vec234.h"
#include "helper.h"
#include "oclhelper.h"
#include "oclimgutil.h"
#include "oclpolyline.h"
#include "oclrect.h"

static cl_device_id g_device;
static cl_context g_context;
static cl_command_queue g_queue;
static int g_inited = 0;

int rdref_init(void) {
  if (g_inited) return 0;
  g_device = simpleGetDevice(0);
  g_context = simpleCreateContext(g_device);
  g_queue = clCreateCommandQueue(g_context, g_device, CL_QUEUE_PROFILING_ENABLE, NULL);
  g_inited = 1;
  return 0;
}

int rdref_sizeof_rect(void) { return (int)sizeof(rect_t); }
int rdref_sizeof_linesegment(void) { return (int)sizeof(linesegment_t); }

/* ------------------------------------------------------------------ rect / vidrect */

typedef struct {
  oclimgutil_t *imgutil;
  oclpolyline_t *polyline;
  struct oclrect_t *rect;
  int iw, ih;
} rdref_rect_t;

void *rdref_rect_open(int iw, int ih) {
  rdref_init();
  rdref_rect_t *h = (rdref_rect_t *)calloc(1, sizeof(*h));
  h->iw = iw; h->ih = ih;
  h->imgutil = init_oclimgutil(g_device, g_context);
  h->polyline = init_oclpolyline(g_device, g_context);
  h->rect = init_oclrect(h->imgutil, h->polyline, g_device, g_context, g_queue, iw, ih);
  return h;
}

void rdref_rect_close(void *hv) {
  rdref_rect_t *h = (rdref_rect_t *)hv;
  dispose_oclrect(h->rect);
  dispose_oclpolyline(h->polyline);
  dispose_oclimgutil(h->imgutil);
  free(h);
}

static int copy_out(rect_t *r, void *out, int max_rects) {
  int n = r->nItems;
  int m = n < max_rects ? n : max_rects;
  if (out && m > 0) memcpy(out, r, (size_t)m * sizeof(rect_t));
  free(r);
  return n;
}

/* returns nItems (count including the header element 0), like rect.cpp:105-107 reads it */
int rdref_rect_execute_once(void *hv, uint8_t *bgr, int ws, double tanAOV, void *out, int max_rects) {
  rdref_rect_t *h = (rdref_rect_t *)hv;
  return copy_out(oclrect_executeOnce(h->rect, bgr, ws, tanAOV), out, max_rects);
}

void rdref_rect_enqueue(void *hv, uint8_t *bgr, int ws) {
  rdref_rect_t *h = (rdref_rect_t *)hv;
  oclrect_enqueueTask(h->rect, bgr, ws);
}

int rdref_rect_poll(void *hv, double tanAOV, void *out, int max_rects) {
  rdref_rect_t *h = (rdref_rect_t *)hv;
  return copy_out(oclrect_pollTask(h->rect, tanAOV), out, max_rects);
}

/* ------------------------------------------------------------------ poly */

/* poly.cpp:74-131 with the literals exposed: strength threshold (500; vidpoly 2000), minerror (1),
 * sizeThre (20; vidpoly 10).  ls_out: 16*N bytes (linesegment_t array, record 0 = header),
 * id_out: N ints (per-pixel segment id, mem0), aux_out: N ints (mem1 as poly.cpp reads it; may be NULL). */
int rdref_poly_run(const uint8_t *bgr, int iw, int ih, int ws, int strengthThre, float minerror, int sizeThre,
                   void *ls_out, int32_t *id_out, int32_t *aux_out) {
  rdref_init();
  const size_t N = (size_t)iw * ih, P = N * sizeof(cl_int);
  if ((size_t)ws * ih > P) return -1;

  oclimgutil_t *oclimgutil = init_oclimgutil(g_device, g_context);
  oclpolyline_t *oclpolyline = init_oclpolyline(g_device, g_context);

  cl_int *zero = (cl_int *)calloc(4, P);
  cl_int *img = (cl_int *)calloc(1, P);
  memcpy(img, bgr, (size_t)ws * ih);

  cl_mem mem[10];
  mem[0] = clCreateBuffer(g_context, CL_MEM_READ_WRITE | CL_MEM_COPY_HOST_PTR, P, img, NULL);
  for (int i = 1; i < 10; i++) mem[i] = clCreateBuffer(g_context, CL_MEM_READ_WRITE | CL_MEM_COPY_HOST_PTR, P, zero, NULL);
  cl_mem memBig = clCreateBuffer(g_context, CL_MEM_READ_WRITE | CL_MEM_COPY_HOST_PTR, P * 4, zero, NULL);
  cl_mem memLS = clCreateBuffer(g_context, CL_MEM_READ_WRITE | CL_MEM_COPY_HOST_PTR, P * 4, zero, NULL);

  ce(clFinish(g_queue));

  cl_command_queue queue = g_queue;
  oclimgutil_convert_plab_bgr(oclimgutil, mem[4], mem[0], iw, ih, ws, queue, NULL);
  oclimgutil_unpack_f_f_f_plab(oclimgutil, mem[1], mem[2], mem[3], mem[4], iw, ih, queue, NULL);
  oclimgutil_iirblur_f_f(oclimgutil, mem[0], mem[1], mem[4], mem[5], 2, iw, ih, queue, NULL);
  oclimgutil_iirblur_f_f(oclimgutil, mem[1], mem[2], mem[4], mem[5], 2, iw, ih, queue, NULL);
  oclimgutil_iirblur_f_f(oclimgutil, mem[2], mem[3], mem[4], mem[5], 2, iw, ih, queue, NULL);
  oclimgutil_pack_plab_f_f_f(oclimgutil, mem[4], mem[0], mem[1], mem[2], iw, ih, queue, NULL);

  oclimgutil_edgevec_f2_f(oclimgutil, memBig, mem[0], iw, ih, queue, NULL);
  oclimgutil_edge_f_plab(oclimgutil, mem[5], mem[4], iw, ih, queue, NULL);
  oclimgutil_thinthres_f_f_f2(oclimgutil, mem[2], mem[5], memBig, iw, ih, queue, NULL);

  oclimgutil_threshold_f_f(oclimgutil, mem[9], mem[2], 0.0, 0.0, 1.0, iw * ih, queue, NULL);
  oclimgutil_cast_i_f(oclimgutil, mem[8], mem[9], 1, iw * ih, queue, NULL);
  oclimgutil_label8x_int_int(oclimgutil, mem[3], mem[8], mem[9], 0, iw, ih, queue, NULL);
  oclimgutil_clear(oclimgutil, mem[4], iw * ih * 4, queue, NULL);
  oclimgutil_calcStrength(oclimgutil, mem[4], mem[2], mem[3], iw, ih, queue, NULL);
  oclimgutil_filterStrength(oclimgutil, mem[3], mem[4], strengthThre, iw, ih, queue, NULL);
  oclimgutil_threshold_i_i(oclimgutil, mem[3], mem[3], 0, 0, 1, iw * ih, queue, NULL);

  oclpolyline_execute(oclpolyline, memLS, iw * ih * 4 * 4, mem[0], mem[3], memBig, mem[4], mem[5], mem[6], mem[7], mem[8], mem[9],
                      minerror, sizeThre, iw, ih, queue, NULL);

  ce(clEnqueueReadBuffer(queue, mem[0], CL_TRUE, 0, P, id_out, 0, NULL, NULL));
  if (aux_out) ce(clEnqueueReadBuffer(queue, mem[1], CL_TRUE, 0, P, aux_out, 0, NULL, NULL));
  ce(clEnqueueReadBuffer(queue, memLS, CL_TRUE, 0, P * 4, ls_out, 0, NULL, NULL));
  ce(clFinish(queue));

  dispose_oclpolyline(oclpolyline);
  dispose_oclimgutil(oclimgutil);
  ce(clReleaseMemObject(memLS));
  ce(clReleaseMemObject(memBig));
  for (int i = 9; i >= 0; i--) ce(clReleaseMemObject(mem[i]));
  free(img);
  free(zero);
  return ((int *)ls_out)[0];
}
