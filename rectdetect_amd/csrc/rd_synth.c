/*
 * Synthetic BGR frame generator: all-integer, seed-driven, identical to the numpy
 * twin in rectdetect_amd/synth.py.  Not part of the detector; it feeds the harness, tests and bench
 * because the reference's OpenCV-based frame sources (rect.cpp:66-74, vidrect.cpp:55-99) are
 * unavailable here.
 *
 *   frame(seed, iw, ih, t): background (40,40,40); K = max(3, round(12*N/2073600)) convex quads,
 *   quad k drawn from an xorshift64* stream keyed by (seed, k): centre anywhere, half sizes in
 *   [ih/16, ih/5], corner jitter <= min(a,b)/3, colour channels in [60,250] (one of them >= 80),
 *   velocity in [-3,3]^2 px/frame with wrap-around of the centre; later quads cover earlier ones;
 *   per-channel noise in [-4,3] from a 32-bit hash of (x, y, t, channel, seed), clamped.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t s; } rng_t;

static uint32_t rng_next(rng_t *r) {
  uint64_t x = r->s;
  x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
  r->s = x;
  return (uint32_t)((x * 0x2545F4914F6CDD1DULL) >> 32);
}

static int rng_int(rng_t *r, int lo, int hi) { return lo + (int)(rng_next(r) % (uint32_t)(hi - lo + 1)); }

static uint32_t noise_hash(uint32_t x, uint32_t y, uint32_t t, uint32_t c, uint32_t seed) {
  uint32_t h = seed ^ (x * 0x9E3779B1u) ^ (y * 0x85EBCA77u) ^ (t * 0xC2B2AE3Du) ^ (c * 0x27D4EB2Fu);
  h ^= h >> 15; h *= 0x2C1B3C6Du;
  h ^= h >> 12; h *= 0x297A2D39u;
  h ^= h >> 15;
  return h;
}

static int pmod(int a, int m) { int r = a % m; return r < 0 ? r + m : r; }

int rd_synth_num_quads(int iw, int ih) {
  long long n = (long long)iw * ih;
  int k = (int)((12 * n + 1036800) / 2073600);
  return k < 3 ? 3 : k;
}

/* Writes the four corners (x0,y0,...,x3,y3) of quad k at time t; returns its colour as 0xRRGGBB. */
uint32_t rd_synth_quad(uint64_t seed, int iw, int ih, int t, int k, int *xy) {
  rng_t r;
  r.s = seed ^ (0x9E3779B97F4A7C15ULL * (uint64_t)(k + 1));
  if (r.s == 0) r.s = 1;
  int cx = rng_int(&r, 0, iw - 1), cy = rng_int(&r, 0, ih - 1);
  int a = rng_int(&r, ih / 16, ih / 5), b = rng_int(&r, ih / 16, ih / 5);
  int j = (a < b ? a : b) / 3;
  static const int sx[4] = { -1, 1, 1, -1 }, sy[4] = { -1, -1, 1, 1 };
  int ox[4], oy[4];
  for (int i = 0; i < 4; i++) {
    ox[i] = sx[i] * a + rng_int(&r, -j, j);
    oy[i] = sy[i] * b + rng_int(&r, -j, j);
  }
  int col[3];
  for (int c = 0; c < 3; c++) col[c] = rng_int(&r, 60, 250);
  if (col[0] < 80 && col[1] < 80 && col[2] < 80) col[k % 3] += 40;
  int vx = rng_int(&r, -3, 3), vy = rng_int(&r, -3, 3);
  int px = pmod(cx + vx * t, iw), py = pmod(cy + vy * t, ih);
  for (int i = 0; i < 4; i++) { xy[2 * i] = px + ox[i]; xy[2 * i + 1] = py + oy[i]; }
  return ((uint32_t)col[2] << 16) | ((uint32_t)col[1] << 8) | (uint32_t)col[0];
}

void rd_synth_frame(uint8_t *bgr, int iw, int ih, int ws, uint64_t seed, int t, int noise) {
  for (int y = 0; y < ih; y++) memset(bgr + (size_t)y * ws, 40, (size_t)iw * 3);

  const int K = rd_synth_num_quads(iw, ih);
  for (int k = 0; k < K; k++) {
    int q[8];
    uint32_t col = rd_synth_quad(seed, iw, ih, t, k, q);
    int x0 = q[0], x1 = q[0], y0 = q[1], y1 = q[1];
    for (int i = 1; i < 4; i++) {
      if (q[2 * i] < x0) x0 = q[2 * i];
      if (q[2 * i] > x1) x1 = q[2 * i];
      if (q[2 * i + 1] < y0) y0 = q[2 * i + 1];
      if (q[2 * i + 1] > y1) y1 = q[2 * i + 1];
    }
    if (x0 < 0) x0 = 0;
    if (y0 < 0) y0 = 0;
    if (x1 > iw - 1) x1 = iw - 1;
    if (y1 > ih - 1) y1 = ih - 1;
    for (int y = y0; y <= y1; y++) {
      for (int x = x0; x <= x1; x++) {
        int in = 1;
        for (int i = 0; i < 4 && in; i++) {
          int ax = q[2 * i], ay = q[2 * i + 1], bx = q[2 * ((i + 1) & 3)], by = q[2 * ((i + 1) & 3) + 1];
          long long cr = (long long)(bx - ax) * (y - ay) - (long long)(by - ay) * (x - ax);
          if (cr < 0) in = 0;
        }
        if (in) {
          uint8_t *p = bgr + (size_t)y * ws + (size_t)x * 3;
          p[0] = (uint8_t)(col & 255); p[1] = (uint8_t)((col >> 8) & 255); p[2] = (uint8_t)((col >> 16) & 255);
        }
      }
    }
  }

  if (noise) {
    const uint32_t s32 = (uint32_t)(seed ^ (seed >> 32));
    for (int y = 0; y < ih; y++) {
      uint8_t *row = bgr + (size_t)y * ws;
      for (int x = 0; x < iw; x++) {
        for (int c = 0; c < 3; c++) {
          int v = row[x * 3 + c] + (int)(noise_hash((uint32_t)x, (uint32_t)y, (uint32_t)t, (uint32_t)c, s32) >> 29) - 4;
          row[x * 3 + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
      }
    }
  }
}
