/* rdpoly - polyline extraction on one still image with the individual operators of oclimgutil.h + oclpolyline_execute, the
 * way the reference's poly.cpp:68-131 chains them (BASELINE.json configs[0]), without OpenCV.
 *
 *   rdpoly <image.ppm|png> [device number] [output.ppm]
 *
 * Prints one line per valid line segment (polyline id, end points) and draws the segments into the output image. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <CL/cl.h>
#include "helper.h"
#include "oclhelper.h"
#include "oclimgutil.h"
#include "oclpolyline.h"
#include "rdimage.h"

int main(int argc, char **argv) {
  if (argc < 2) {
    fprintf(stderr, "Usage : %s <image file name (.ppm or .png)> [device number] [output file name (.ppm)]\n\nAvailable devices :\n", argv[0]);
    simpleGetDevice(-1);
    return 1;
  }
  cl_device_id device = simpleGetDevice(argc >= 3 ? atoi(argv[2]) : 0);
  printf("%s\n", getDeviceName(device));
  cl_context context = simpleCreateContext(device);
  cl_command_queue queue = clCreateCommandQueue(context, device, 0, NULL);
  rdimage img;
  if (rdimage_load(argv[1], &img) != 0) return 1;
  const int iw = img.iw, ih = img.ih, ws = img.ws;
  const size_t plane = (size_t)iw * ih * sizeof(cl_int);

  oclimgutil_t *iu = init_oclimgutil(device, context);
  oclpolyline_t *pl = init_oclpolyline(device, context);

  /* ten int-sized planes (plane 0 starts as the BGR frame), one 4x plane for the direction field, one for the records */
  cl_mem m[10], big, ls;
  uint8_t *frame = (uint8_t *)calloc(plane, 1);
  memcpy(frame, img.bgr, (size_t)ws * ih);
  m[0] = clCreateBuffer(context, CL_MEM_READ_WRITE | CL_MEM_COPY_HOST_PTR, plane, frame, NULL);
  for (int k = 1; k < 10; k++) m[k] = clCreateBuffer(context, CL_MEM_READ_WRITE, plane, NULL, NULL);   /* zero-filled */
  big = clCreateBuffer(context, CL_MEM_READ_WRITE, plane * 4, NULL, NULL);
  ls = clCreateBuffer(context, CL_MEM_READ_WRITE, plane * 4, NULL, NULL);
  ce(clFinish(queue));

  /* colour -> blurred Lab -> gradient direction, strength, thinning */
  oclimgutil_convert_plab_bgr(iu, m[4], m[0], iw, ih, ws, queue, NULL);
  oclimgutil_unpack_f_f_f_plab(iu, m[1], m[2], m[3], m[4], iw, ih, queue, NULL);
  for (int k = 0; k < 3; k++) oclimgutil_iirblur_f_f(iu, m[k], m[k + 1], m[4], m[5], 2, iw, ih, queue, NULL);
  oclimgutil_pack_plab_f_f_f(iu, m[4], m[0], m[1], m[2], iw, ih, queue, NULL);
  oclimgutil_edgevec_f2_f(iu, big, m[0], iw, ih, queue, NULL);
  oclimgutil_edge_f_plab(iu, m[5], m[4], iw, ih, queue, NULL);
  oclimgutil_thinthres_f_f_f2(iu, m[2], m[5], big, iw, ih, queue, NULL);
  /* connected edges with enough accumulated strength -> 0/1 mask */
  oclimgutil_threshold_f_f(iu, m[9], m[2], 0.0f, 0.0f, 1.0f, iw * ih, queue, NULL);
  oclimgutil_cast_i_f(iu, m[8], m[9], 1.0f, iw * ih, queue, NULL);
  oclimgutil_label8x_int_int(iu, m[3], m[8], m[9], 0, iw, ih, queue, NULL);
  oclimgutil_clear(iu, m[4], iw * ih * 4, queue, NULL);
  oclimgutil_calcStrength(iu, m[4], m[2], m[3], iw, ih, queue, NULL);
  oclimgutil_filterStrength(iu, m[3], m[4], 500, iw, ih, queue, NULL);
  oclimgutil_threshold_i_i(iu, m[3], m[3], 0, 0, 1, iw * ih, queue, NULL);
  /* polylines */
  oclpolyline_execute(pl, ls, iw * ih * 4 * 4, m[0], m[3], big, m[4], m[5], m[6], m[7], m[8], m[9], 1.0f, 20, iw, ih, queue, NULL);

  linesegment_t hdr;
  ce(clEnqueueReadBuffer(queue, ls, CL_TRUE, 0, sizeof(hdr), &hdr, 0, NULL, NULL));
  const int n = *(int *)&hdr;
  linesegment_t *seg = (linesegment_t *)malloc((size_t)(n + 1) * sizeof(linesegment_t));
  ce(clEnqueueReadBuffer(queue, ls, CL_TRUE, 0, (size_t)(n + 1) * sizeof(linesegment_t), seg, 0, NULL, NULL));
  int valid = 0;
  for (int i = 1; i <= n; i++) valid += seg[i].polyid != 0;
  printf("%d record(s), %d segment(s)\n", n, valid);
  for (int i = 1; i <= n; i++) {
    if (seg[i].polyid == 0) continue;
    printf("segment %d polyline %d (%.3f, %.3f) - (%.3f, %.3f)\n", i, seg[i].polyid, seg[i].x0, seg[i].y0, seg[i].x1, seg[i].y1);
    const unsigned c = (unsigned)seg[i].polyid * 2654435761u;
    rdimage_line(&img, seg[i].x0, seg[i].y0, seg[i].x1, seg[i].y1, 64 + (c >> 8 & 191), 64 + (c >> 16 & 191), 64 + (c >> 24 & 191), 2);
  }
  rdimage_save_ppm(argc >= 4 ? argv[3] : "output.ppm", &img);

  free(seg); free(frame);
  for (int k = 0; k < 10; k++) ce(clReleaseMemObject(m[k]));
  ce(clReleaseMemObject(big)); ce(clReleaseMemObject(ls));
  dispose_oclpolyline(pl);
  dispose_oclimgutil(iu);
  ce(clReleaseCommandQueue(queue));
  ce(clReleaseContext(context));
  rdimage_free(&img);
  return 0;
}
