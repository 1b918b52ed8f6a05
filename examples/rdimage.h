/* Minimal image I/O for the OpenCV-free example programs: binary PPM (P6) in and out, 8-bit RGB/RGBA non-interlaced PNG in
 * (zlib inflate + the five scanline filters).  Pixels are handed over as BGR, 3 bytes per pixel, rows `ws` bytes apart -
 * the layout the detector API takes (what cv::Mat::data / step are in the reference's programs). */
#ifndef RDIMAGE_H
#define RDIMAGE_H
#include <stdint.h>

typedef struct { int iw, ih, ws; uint8_t *bgr; } rdimage;

/* returns 0 on success; on failure prints the reason to stderr and returns non-zero */
int rdimage_load(const char *path, rdimage *img);
int rdimage_save_ppm(const char *path, const rdimage *img);
void rdimage_free(rdimage *img);
/* 1-pixel (or thicker) line from (x0, y0) to (x1, y1), clipped to the image */
void rdimage_line(rdimage *img, double x0, double y0, double x1, double y1, int r, int g, int b, int thickness);
#endif
