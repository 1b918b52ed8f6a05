#include "rdimage.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

static int fail(const char *path, const char *why) { fprintf(stderr, "%s: %s\n", path, why); return -1; }

static int ppm_token(FILE *f, int *out) {   /* next unsigned integer of a PNM header, skipping white space and # comments */
  int c;
  for (;;) {
    c = fgetc(f);
    if (c == '#') { while (c != '\n' && c != EOF) c = fgetc(f); continue; }
    if (c == EOF) return -1;
    if (c > ' ') break;
  }
  int v = 0;
  while (c >= '0' && c <= '9') { v = v * 10 + (c - '0'); c = fgetc(f); }
  *out = v;
  return 0;   /* the single white-space byte after the token has been consumed */
}

static int load_ppm(FILE *f, const char *path, rdimage *img) {
  int w, h, maxv;
  if (ppm_token(f, &w) || ppm_token(f, &h) || ppm_token(f, &maxv)) return fail(path, "truncated PPM header");
  if (w < 1 || h < 1 || maxv != 255) return fail(path, "only 8-bit P6 images are supported");
  if (w > 32768 || h > 32768 || (long long)w * h >= (1 << 25)) return fail(path, "PPM dimensions out of range (frames must be below 2^25 pixels)");
  img->iw = w; img->ih = h; img->ws = w * 3;
  img->bgr = (uint8_t *)malloc((size_t)img->ws * h);
  uint8_t *row = (uint8_t *)malloc((size_t)w * 3);
  if (!img->bgr || !row) { free(img->bgr); img->bgr = NULL; free(row); return fail(path, "out of memory"); }
  for (int y = 0; y < h; y++) {
    if (fread(row, 3, (size_t)w, f) != (size_t)w) { free(row); return fail(path, "truncated PPM data"); }
    uint8_t *o = img->bgr + (size_t)y * img->ws;
    for (int x = 0; x < w; x++) { o[x * 3] = row[x * 3 + 2]; o[x * 3 + 1] = row[x * 3 + 1]; o[x * 3 + 2] = row[x * 3]; }
  }
  free(row);
  return 0;
}

static uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

static int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

static int load_png(FILE *f, const char *path, rdimage *img) {
  uint8_t hd[8];
  int w = 0, h = 0, bpp = 0;
  uint8_t *z = NULL; size_t zn = 0, zcap = 0;
  for (;;) {
    if (fread(hd, 1, 8, f) != 8) { free(z); return fail(path, "truncated PNG"); }
    const uint32_t len = be32(hd);
    if (len > (1u << 28)) { free(z); return fail(path, "PNG chunk too large"); }      /* (the detector takes frames below 2^25 pixels) */
    uint8_t *data = (uint8_t *)malloc(len ? len : 1);
    if (!data) { free(z); return fail(path, "out of memory"); }
    if (fread(data, 1, len, f) != len || fread(hd + 0, 1, 4, f) != 4 /* CRC, not checked */) { free(data); free(z); return fail(path, "truncated PNG chunk"); }
    if (!memcmp(hd + 4, "IHDR", 4)) {
      if (len < 13) { free(data); free(z); return fail(path, "PNG header chunk too short"); }
      const uint32_t uw = be32(data), uh = be32(data + 4);
      if (uw < 1 || uh < 1 || uw > 32768 || uh > 32768 || (uint64_t)uw * uh >= (1u << 25)) { free(data); free(z); return fail(path, "PNG dimensions out of range (frames must be below 2^25 pixels)"); }
      w = (int)uw; h = (int)uh;
      if (data[8] != 8 || (data[9] != 2 && data[9] != 6) || data[12] != 0) { free(data); free(z); return fail(path, "only 8-bit RGB / RGBA non-interlaced PNG is supported"); }
      bpp = data[9] == 2 ? 3 : 4;
    } else if (!memcmp(hd + 4, "IDAT", 4)) {
      if (zn + len > zcap) {
        zcap = (zn + len) * 2;
        uint8_t *nz = (uint8_t *)realloc(z, zcap);
        if (!nz) { free(data); free(z); return fail(path, "out of memory"); }
        z = nz;
      }
      memcpy(z + zn, data, len); zn += len;
    } else if (!memcmp(hd + 4, "IEND", 4)) { free(data); break; }
    free(data);
  }
  if (w < 1 || h < 1 || !bpp) { free(z); return fail(path, "PNG without header"); }
  const size_t stride = (size_t)w * bpp + 1;
  uLongf rawn = (uLongf)(stride * h);
  uint8_t *raw = (uint8_t *)malloc(rawn);
  if (!raw) { free(z); return fail(path, "out of memory"); }
  if (uncompress(raw, &rawn, z, (uLong)zn) != Z_OK || rawn != stride * h) { free(raw); free(z); return fail(path, "PNG data does not inflate"); }
  free(z);
  img->iw = w; img->ih = h; img->ws = w * 3;
  img->bgr = (uint8_t *)malloc((size_t)img->ws * h);
  uint8_t *prev = (uint8_t *)calloc(stride, 1);
  if (!img->bgr || !prev) { free(img->bgr); img->bgr = NULL; free(prev); free(raw); return fail(path, "out of memory"); }
  for (int y = 0; y < h; y++) {
    uint8_t *line = raw + (size_t)y * stride + 1;
    const int ft = line[-1];
    for (int i = 0; i < w * bpp; i++) {
      const int a = i >= bpp ? line[i - bpp] : 0, b = prev[i + 1], c = i >= bpp ? prev[i + 1 - bpp] : 0;
      int v = line[i];
      switch (ft) { case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) / 2; break; case 4: v += paeth(a, b, c); break; default: break; }
      line[i] = (uint8_t)v;
    }
    memcpy(prev + 1, line, (size_t)w * bpp);
    uint8_t *o = img->bgr + (size_t)y * img->ws;
    for (int x = 0; x < w; x++) { o[x * 3] = line[x * bpp + 2]; o[x * 3 + 1] = line[x * bpp + 1]; o[x * 3 + 2] = line[x * bpp]; }
  }
  free(prev); free(raw);
  return 0;
}

int rdimage_load(const char *path, rdimage *img) {
  memset(img, 0, sizeof(*img));
  FILE *f = fopen(path, "rb");
  if (!f) return fail(path, "cannot open");
  uint8_t sig[8];
  int rc;
  if (fread(sig, 1, 2, f) == 2 && sig[0] == 'P' && sig[1] == '6') rc = load_ppm(f, path, img);
  else if (fread(sig + 2, 1, 6, f) == 6 && !memcmp(sig, "\x89PNG\r\n\x1a\n", 8)) rc = load_png(f, path, img);
  else rc = fail(path, "neither a P6 PPM nor a PNG file");
  fclose(f);
  return rc;
}

int rdimage_save_ppm(const char *path, const rdimage *img) {
  FILE *f = fopen(path, "wb");
  if (!f) return fail(path, "cannot create");
  fprintf(f, "P6\n%d %d\n255\n", img->iw, img->ih);
  for (int y = 0; y < img->ih; y++)
    for (int x = 0; x < img->iw; x++) {
      const uint8_t *p = img->bgr + (size_t)y * img->ws + x * 3;
      fputc(p[2], f); fputc(p[1], f); fputc(p[0], f);
    }
  fclose(f);
  return 0;
}

void rdimage_free(rdimage *img) { free(img->bgr); img->bgr = NULL; }

void rdimage_line(rdimage *img, double x0, double y0, double x1, double y1, int r, int g, int b, int thickness) {
  const double dx = x1 - x0, dy = y1 - y0;
  const int n = (int)ceil(fmax(fabs(dx), fabs(dy))) + 1;
  for (int i = 0; i <= n; i++) {
    const double t = n ? (double)i / n : 0.0;
    const int cx = (int)lrint(x0 + dx * t), cy = (int)lrint(y0 + dy * t);
    for (int oy = -(thickness - 1) / 2; oy <= thickness / 2; oy++)
      for (int ox = -(thickness - 1) / 2; ox <= thickness / 2; ox++) {
        const int x = cx + ox, y = cy + oy;
        if (x < 0 || y < 0 || x >= img->iw || y >= img->ih) continue;
        uint8_t *p = img->bgr + (size_t)y * img->ws + x * 3;
        p[0] = (uint8_t)b; p[1] = (uint8_t)g; p[2] = (uint8_t)r;
      }
  }
}
