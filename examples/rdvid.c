/* rdvid - the steady-state loop of the reference's vidrect.cpp:159-205 (enqueue frame k, poll frame k-1) through the
 * reference's C API, without OpenCV: frames come from the synthetic stream generator of this library (rd_synth_frame) or
 * from numbered PPM/PNG files.
 *
 *   rdvid <width> <height> <frames> [device number] [angle of view in degrees]        synthetic stream
 *   rdvid <printf pattern, e.g. frame%04d.ppm> <first> <count> [device number] [aov]   image sequence
 *
 * Prints the number of rectangles per frame and, once per second, the frame rate (as vidrect.cpp does when writing a file). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <CL/cl.h>
#include "helper.h"
#include "oclhelper.h"
#include "oclimgutil.h"
#include "oclpolyline.h"
#include "vec234.h"
#include "oclrect.h"
#include "rectdetect_hip.h"
#include "rdimage.h"

int main(int argc, char **argv) {
  if (argc < 4) {
    fprintf(stderr, "Usage : %s <width> <height> <frames> [device] [aov]\n        %s <pattern like f%%04d.ppm> <first> <count> [device] [aov]\n", argv[0], argv[0]);
    return 1;
  }
  const int from_files = strchr(argv[1], '%') != NULL;
  const int a2 = atoi(argv[2]), a3 = atoi(argv[3]);
  const int did = argc >= 5 ? atoi(argv[4]) : 0;
  const double aov = argc >= 6 ? atof(argv[5]) : 72.0;
  const double tanAOV = tan(aov / 2 / 180.0 * M_PI);

  rdimage page[2] = { { 0, 0, 0, NULL }, { 0, 0, 0, NULL } };   /* two pages, like the reference's img[2]: a frame stays valid until its poll */
  int iw, ih;
  const int nframes = a3;
  char name[1024];
  if (from_files) {
    snprintf(name, sizeof(name), argv[1], a2);
    if (rdimage_load(name, &page[0]) != 0) return 1;
    iw = page[0].iw; ih = page[0].ih;
  } else {
    iw = atoi(argv[1]); ih = a2;
    for (int k = 0; k < 2; k++) { page[k].iw = iw; page[k].ih = ih; page[k].ws = iw * 3; page[k].bgr = (uint8_t *)malloc((size_t)iw * 3 * ih); }
  }

  cl_device_id device = simpleGetDevice(did);
  printf("%s\n", getDeviceName(device));
  cl_context context = simpleCreateContext(device);
  cl_command_queue queue = clCreateCommandQueue(context, device, 0, NULL);
  struct oclimgutil_t *iu = init_oclimgutil(device, context);
  struct oclpolyline_t *pl = init_oclpolyline(device, context);
  struct oclrect_t *rc = init_oclrect(iu, pl, device, context, queue, iw, ih);

  uint64_t tm = currentTimeMillis();
  int last = 0, pending = 0, polled = 0;
  for (int n = 0; n <= nframes; n++) {
    if (n < nframes) {
      rdimage *pg = &page[n & 1];
      if (from_files) {
        if (n > 0) { rdimage_free(pg); snprintf(name, sizeof(name), argv[1], a2 + n); if (rdimage_load(name, pg) != 0) break; }
        if (pg->iw != iw || pg->ih != ih) { fprintf(stderr, "%s: frame size changed\n", name); break; }
      } else rd_synth_frame(pg->bgr, iw, ih, pg->ws, 0x5EED0000ull, n, 1);
      oclrect_enqueueTask(rc, pg->bgr, pg->ws);
      pending++;
    }
    if (pending == 2 || (n == nframes && pending == 1)) {
      rect_t *ret = oclrect_pollTask(rc, tanAOV);
      printf("frame %d: %d rectangle(s)\n", polled++, ret->nItems - 1);
      free(ret);
      pending--;
    }
    const uint64_t t = currentTimeMillis();
    if (t - tm > 1000) { printf("%.3g fps\n", 1000.0 * (n - last) / (double)(t - tm)); tm = t; last = n; }
  }

  dispose_oclrect(rc);
  dispose_oclpolyline(pl);
  dispose_oclimgutil(iu);
  ce(clReleaseCommandQueue(queue));
  ce(clReleaseContext(context));
  for (int k = 0; k < 2; k++) rdimage_free(&page[k]);
  return 0;
}
