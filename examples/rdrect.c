/* rdrect - rectangle detection on one still image through the reference's C API (oclhelper.h / oclimgutil.h /
 * oclpolyline.h / oclrect.h), without OpenCV: the call sequence of the reference's rect.cpp:47-138 with the image
 * decoded by rdimage.c (P6 PPM or 8-bit PNG) and the result drawn into a PPM.
 *
 *   rdrect <image.ppm|png> [device number] [output.ppm]
 *
 * Prints one line per rectangle: status, score, the four image-plane corners.  Links against librectdetect_hip.so only. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <CL/cl.h>
#include "helper.h"
#include "oclhelper.h"
#include "oclimgutil.h"
#include "oclpolyline.h"
#include "vec234.h"
#include "oclrect.h"
#include "rdimage.h"

static void outline(rdimage *img, const rect_t *q, int r, int g, int b, int thickness) {
  for (int k = 0; k < 4; k++) rdimage_line(img, q->c2[k].a[0], q->c2[k].a[1], q->c2[(k + 1) & 3].a[0], q->c2[(k + 1) & 3].a[1], r, g, b, thickness);
  rdimage_line(img, q->c2[0].a[0], q->c2[0].a[1], q->c2[2].a[0], q->c2[2].a[1], r, g, b, 1);   /* diagonals, as the reference draws them */
  rdimage_line(img, q->c2[1].a[0], q->c2[1].a[1], q->c2[3].a[0], q->c2[3].a[1], r, g, b, 1);
}

int main(int argc, char **argv) {
  if (argc < 2) {
    fprintf(stderr, "Usage : %s <image file name (.ppm or .png)> [device number] [output file name (.ppm)]\n\nAvailable devices :\n", argv[0]);
    simpleGetDevice(-1);
    return 1;
  }
  const int did = argc >= 3 ? atoi(argv[2]) : 0;
  cl_device_id device = simpleGetDevice(did);
  printf("%s\n", getDeviceName(device));
  cl_context context = simpleCreateContext(device);
  cl_command_queue queue = clCreateCommandQueue(context, device, CL_QUEUE_PROFILING_ENABLE, NULL);

  rdimage img;
  if (rdimage_load(argv[1], &img) != 0) return 1;

  struct oclimgutil_t *iu = init_oclimgutil(device, context);
  struct oclpolyline_t *pl = init_oclpolyline(device, context);
  struct oclrect_t *rc = init_oclrect(iu, pl, device, context, queue, img.iw, img.ih);

  const double tanAOV = tan(72.0 / 2 / 180.0 * M_PI);
  if (loadPlan("plan.txt", device) != 0) savePlan("plan.txt", device);   /* kept for compatibility: launch shapes are fixed in this build */

  rect_t *ret = oclrect_executeOnce(rc, img.bgr, img.ws, tanAOV);
  printf("%d rectangle(s)\n", ret->nItems - 1);
  for (int i = 1; i < ret->nItems; i++) {   /* element 0 is the header */
    const rect_t *q = &ret[i];
    printf("status %u value %.6g corners", q->status, q->value);
    for (int k = 0; k < 4; k++) printf(" (%.3f, %.3f)", q->c2[k].a[0], q->c2[k].a[1]);
    printf("\n");
    switch (q->status) {
    case 0: case 2: outline(&img, q, 255, 0, 0, 1); break;
    case 1: outline(&img, q, 0, 200, 255, 2); break;
    default: outline(&img, q, 0, 0, 255, 2); break;
    }
  }
  free(ret);
  rdimage_save_ppm(argc >= 4 ? argv[3] : "output.ppm", &img);

  dispose_oclrect(rc);
  dispose_oclpolyline(pl);
  dispose_oclimgutil(iu);
  ce(clReleaseCommandQueue(queue));
  ce(clReleaseContext(context));
  rdimage_free(&img);
  return 0;
}
