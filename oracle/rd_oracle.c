/*
 * TEST INFRASTRUCTURE ONLY.  CPU restatement ("oracle") of rectdetect's per-frame device
 * stages, in plain C, one function per stage.  It is the checker for the HIP kernels and the
 * "port" CPU baseline of bench.py; the product library never calls, links or loads it.
 *
 * Parity pin: every stage here is compared (tests/test_oracle_vs_ref.py) against the reference's own
 * kernels + host C executed on the serial OpenCL shim (oracle/_ref, built from /root/reference).
 * The reference ships no tests or golden vectors of its own.  Round 5: the same reference objects also link against the
 * system's real OpenCL loader (oracle/_ref/librdref_ocl.so); on the GPU box the vendor's compiler and the MI355X reproduce all operator
 * goldens bit for bit under this file's arithmetic contract (tools/ref_ops_on_opencl.py), and in whole frames every plane of
 * this file up to the merge masks (16 planes, observed launch by launch: refshim/rdcl_observe.c, tools/ref_stages_on_opencl.py); the stages
 * from the in-place region merge on depend on the device's work-item order and are pinned by the stand-in's raster order only - parity
 * "partial" by the tier's strict rule.
 *
 * Canonical semantics: work-items in raster order; labelling stages are run to
 * convergence (label = smallest pixel index of the component); no FMA contraction; OpenCL
 * builtins as defined in oracle/refshim/rdcl_builtins.c.
 *
 * Citations are file:line into the reference (oclimgutil.cl = "iu", oclrect.cl = "rc",
 * oclrect.c = "rh", oclimgutil.c = "ih").
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "rd_oracle.h"

/* ------------------------------------------------------------------ small helpers */

static inline int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
static inline uint32_t clampu(uint32_t x, uint32_t lo, uint32_t hi) { return x < lo ? lo : (x > hi ? hi : x); }
static inline int mirror1(int x, int n) { return clampi(x, -x, 2 * n - 2 - x); }          /* iu:47 */
static inline int mirror2(int x, int y, int iw, int ih) { return mirror1(x, iw) + mirror1(y, ih) * iw; } /* iu:41 */
static inline int wrap1(int x, int n) { x = x < 0 ? x + n : x; return x >= n ? x - n : x; } /* iu:51 */

static inline uint32_t floor_u32(float f) {               /* convert_uint_rtn(float) */
  float g = floorf(f);
  if (!(g > 0.0f)) return 0u;
  if (g >= 4294967296.0f) return 0xffffffffu;
  return (uint32_t)g;
}

/* iu:28-34 */
static inline uint32_t pack_lab(float L, float a, float b) {
  uint32_t r = clampu(floor_u32(b * 1024), 0u, 1023u);
  r = (r << 10) | clampu(floor_u32(a * 1024), 0u, 1023u);
  r = (r << 12) | clampu(floor_u32(L * 4096), 0u, 4095u);
  return r;
}

/* iu:36-39 */
static inline void unpack_lab(uint32_t p, float *L, float *a, float *b) {
  *L = (float)(int)(p & 4095u) * (1.0f / 4096) + 0.5f / 4096;
  *a = (float)(int)((p >> 12) & 1023u) * (1.0f / 1024) + 0.5f / 1024;
  *b = (float)(int)((p >> 22) & 1023u) * (1.0f / 1024) + 0.5f / 1024;
}

/* ------------------------------------------------------------------ colour LUTs */
/* Closed forms that reproduce iu:661-898 entry for entry (checked in tests):
 *   s2l[i]   = floor(32768 * srgb_to_linear(i/255)),                    i = 0..260
 *   cfunc[i] = rint(65536 * f(i/1024) - 9039),  f = CIE Lab f(t),        i = 0..1030
 *   cfunc2[i]= rint(65536/127.5 * (116 f(i/1024) - 16))                                  */
static uint16_t lut_s2l[261], lut_cf[1031], lut_cf2[1031], lut_l2s[1024];   /* l2s (iu:697-762) = clamp(floor(256 * linear_to_srgb(i/1023) + 0.5), 0, 255) */
static int luts_ready = 0;

static void init_luts(void) {
  if (luts_ready) return;
  for (int i = 0; i < 261; i++) {
    double c = i / 255.0;
    double lin = c <= 0.04045 ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4);
    lut_s2l[i] = (uint16_t)floor(32768 * lin);
  }
  for (int i = 0; i < 1031; i++) {
    double t = i / 1024.0;
    double f = t > 0.008856 ? pow(t, 1.0 / 3) : 7.787 * t + 16.0 / 116;
    lut_cf[i] = (uint16_t)rint(65536 * f - 9039);
    lut_cf2[i] = (uint16_t)rint(65536 / 127.5 * (116 * f - 16));
  }
  for (int i = 0; i < 1024; i++) {
    double c = i / 1023.0;
    double v = c <= 0.0031308 ? 12.92 * c : 1.055 * pow(c, 1 / 2.4) - 0.055;
    int q = (int)floor(256 * v + 0.5);
    lut_l2s[i] = (uint16_t)(q < 0 ? 0 : q > 255 ? 255 : q);
  }
  luts_ready = 1;
}

const uint16_t *rdo_lut(int which, int *n) {
  init_luts();
  if (which == 0) { *n = 261; return lut_s2l; }
  if (which == 1) { *n = 1031; return lut_cf; }
  if (which == 3) { *n = 1024; return lut_l2s; }
  *n = 1031; return lut_cf2;
}

/* iu:106-134: sRGB -> packed Lab in integer arithmetic.  Matrix entries are the values of
 * (int)(m * 16384 + 0.5f) for the sRGB->XYZ matrix, 34476 = (int)(32768/0.950456f + 0.5f),
 * 30097 = (int)(32768/1.088754f + 0.5f). */
static inline int lerp_lut(const uint16_t *t, int c) { return t[c >> 8] * (256 - (c & 255)) + t[(c >> 8) + 1] * (c & 255); }

static inline uint32_t srgb_to_plab(int B, int G, int R) {
  int ir = lut_s2l[R], ig = lut_s2l[G], ib = lut_s2l[B];
  int cx = (((ir * 6758 + ig * 5859 + ib * 2956 + (1 << 14)) >> 15) * 34476 + (1 << 10)) >> 11;
  int cy = ((ir * 3484 + ig * 11717 + ib * 1182) + (1 << 10)) >> 11;
  int cz = (((ir * 317 + ig * 1953 + ib * 15569 + (1 << 14)) >> 15) * 30097 + (1 << 10)) >> 11;
  int cl = ((lerp_lut(lut_cf2, cy) >> 12) + 1) >> 1;
  int fx = lerp_lut(lut_cf, cx), fy = lerp_lut(lut_cf, cy), fz = lerp_lut(lut_cf, cz);
  int fxy = (fx - fy + (1 << 7)) >> 8, fyz = (fy - fz + (1 << 7)) >> 8;
  int ca = (fxy * 8031 + (134744072 + (1 << 17))) >> 18;
  int cb = (fyz * 3213 + (134744072 + (1 << 17))) >> 18;
  uint32_t r = clampu((uint32_t)cb, 0u, 1023u);
  r = (r << 10) | clampu((uint32_t)ca, 0u, 1023u);
  r = (r << 12) | clampu((uint32_t)cl, 0u, 4095u);
  return r;
}

/* iu:256-262 */
void rdo_bgr2plab(uint32_t *out, const uint8_t *bgr, int iw, int ih, int ws) {
  init_luts();
  for (int y = 0; y < ih; y++)
    for (int x = 0; x < iw; x++) {
      const uint8_t *p = bgr + (size_t)y * ws + x * 3;
      out[y * iw + x] = srgb_to_plab(p[0], p[1], p[2]);
    }
}

/* iu:333-342 */
void rdo_unpack_plab(float *L, float *a, float *b, const uint32_t *in, int n) {
  for (int i = 0; i < n; i++) unpack_lab(in[i], &L[i], &a[i], &b[i]);
}

/* iu:325-331 */
void rdo_pack_plab(uint32_t *out, const float *L, const float *a, const float *b, int n) {
  for (int i = 0; i < n; i++) out[i] = pack_lab(L[i], a[i], b[i]);
}

/* ------------------------------------------------------------------ IIR Gaussian, sigma = 1 */
/* iircoef[2] (iu:915-921): 8 feed-forward and 7 feedback taps.  The apps only ever pass r = 2. */
static const float IIR[15] = {
  0.3989422804f, 0.1414542400f, -0.0030406818f, -0.0041116157f, 0.0006696623f, 0.0000498707f, -0.0000449761f, -0.0000051528f,
  0.2519574622f, -0.0098627835f, -0.0067013653f, 0.0012572396f, 0.0000481394f, -0.0000097781f, 0.0000006462f,
};
#define IIR_R 2
#define IIR_WARM (IIR_R + 1 + 8)
#include "rd_iircoef.h"     /* every radius (iu:900-1125), generated by tools/gen_iircoef.py; row 2 = IIR above */

/* One causal sweep over a line of n samples with stride `st` (iu:542-559 and the three siblings):
 * positions start .. end in direction dir, warm-up samples mirrored, results written at wrapped
 * positions so that the final value at each in-range position is the true recurrence value. */
static void iir_sweep(float *dst, const float *src, int n, int st, int dir, const float *C, int warm) {
  float iv[8] = { 0 }, tv[8] = { 0 };
  int x = dir > 0 ? -warm : n + warm;
  for (;;) {
    if (dir > 0 ? x >= n : x < 0) break;
    iv[0] = src[mirror1(x, n) * st];
    float d = iv[0] * C[0];
    d += C[1] * iv[1] + C[2] * iv[2] + C[3] * iv[3] + C[4] * iv[4] + C[5] * iv[5] + C[6] * iv[6] + C[7] * iv[7];
    d += C[8] * tv[0] + C[9] * tv[1] + C[10] * tv[2] + C[11] * tv[3] + C[12] * tv[4] + C[13] * tv[5] + C[14] * tv[6];
    dst[wrap1(x, n) * st] = d;
    for (int k = 7; k > 0; k--) { iv[k] = iv[k - 1]; tv[k] = tv[k - 1]; }
    tv[0] = d;
    x += dir;
  }
}

/* ih:248-273 + iu:542-637: out = vertical(horizontal(in)); each direction = causal + anti-causal - c0*in */
static void iirblur_with(float *out, const float *in, int iw, int ih, const float *C, int warm) {
  const int N = iw * ih;
  float *t0 = (float *)malloc(sizeof(float) * N), *t1 = (float *)malloc(sizeof(float) * N);
  for (int y = 0; y < ih; y++) {
    iir_sweep(t0 + y * iw, in + y * iw, iw, 1, +1, C, warm);
    iir_sweep(t1 + y * iw, in + y * iw, iw, 1, -1, C, warm);
  }
  for (int i = 0; i < N; i++) out[i] = t1[i] + t0[i] - in[i] * C[0];
  for (int x = 0; x < iw; x++) {
    iir_sweep(t0 + x, out + x, ih, iw, +1, C, warm);
    iir_sweep(t1 + x, out + x, ih, iw, -1, C, warm);
  }
  for (int i = 0; i < N; i++) out[i] = t1[i] + t0[i] - out[i] * C[0];
  free(t0); free(t1);
}

void rdo_iirblur(float *out, const float *in, int iw, int ih) { iirblur_with(out, in, iw, ih, IIR, IIR_R + 1 + 8); }

/* any radius r = 0..31 (sigma = (r + 1) / 3); lines must be at least r + 11 samples long (shorter ones make the reference's
 * mirrored warm-up read outside the plane).  Returns 0, or -1 for arguments outside that domain. */
int rdo_iirblur_r(float *out, const float *in, int iw, int ih, int r) {
  if (r < 0 || r >= RD_IIRCOEF_NR || iw < r + 11 || ih < r + 11) return -1;
  iirblur_with(out, in, iw, ih, rd_iircoef[r], r + 1 + 8);
  return 0;
}
const float *rdo_iircoef(int r) { return r == -1 ? IIR : rd_iircoef[r]; }   /* (-1: the constants of the sigma = 1 path) */

/* ------------------------------------------------------------------ gradient direction, strength, NMS */

/* iu:346-352: the 5x5 kernel; double literals narrowed to float as the OpenCL compiler does */
static const float V5C[25] = {
  (float)-4.667, (float)-4.083, (float)0.000, (float)4.083, (float)4.667,
  (float)-10.024, (float)-0.963, (float)0.000, (float)0.963, (float)10.024,
  (float)-14.120, (float)3.622, (float)0.000, (float)-3.622, (float)14.120,
  (float)-10.024, (float)-0.963, (float)0.000, (float)0.963, (float)10.024,
  (float)-4.667, (float)-4.083, (float)0.000, (float)4.083, (float)4.667,
};

/* iu:395-420 */
void rdo_edgevec(float *vxy, const float *in, int iw, int ih) {
  for (int y = 0; y < ih; y++)
    for (int x = 0; x < iw; x++) {
      float vx = 0, vy = 0;
      for (int yy = -2; yy <= 2; yy++)
        for (int xx = -2; xx <= 2; xx++) {
          float s = in[mirror2(x + xx, y + yy, iw, ih)];
          vx += V5C[(xx + 2) + (yy + 2) * 5] * s;
          vy += V5C[(yy + 2) + (xx + 2) * 5] * s;
        }
      float len = vx * vx + vy * vy;
      if ((double)len > 1e-10) {
        len = 1.0f / sqrtf(len);
        vx *= len; vy *= len;
      } else {
        vx = vy = 0.70710678118f;
      }
      vxy[2 * (y * iw + x)] = vx;
      vxy[2 * (y * iw + x) + 1] = vy;
    }
}

/* iu:422-437 */
void rdo_edge_plab(float *out, const uint32_t *in, int iw, int ih) {
  for (int y = 0; y < ih; y++)
    for (int x = 0; x < iw; x++) {
      float n[3], s[3], w[3], e[3], nw[3], ne[3], sw[3], se[3], sum[3];
      unpack_lab(in[mirror2(x, y - 1, iw, ih)], &n[0], &n[1], &n[2]);
      unpack_lab(in[mirror2(x, y + 1, iw, ih)], &s[0], &s[1], &s[2]);
      unpack_lab(in[mirror2(x - 1, y, iw, ih)], &w[0], &w[1], &w[2]);
      unpack_lab(in[mirror2(x + 1, y, iw, ih)], &e[0], &e[1], &e[2]);
      unpack_lab(in[mirror2(x - 1, y - 1, iw, ih)], &nw[0], &nw[1], &nw[2]);
      unpack_lab(in[mirror2(x + 1, y - 1, iw, ih)], &ne[0], &ne[1], &ne[2]);
      unpack_lab(in[mirror2(x - 1, y + 1, iw, ih)], &sw[0], &sw[1], &sw[2]);
      unpack_lab(in[mirror2(x + 1, y + 1, iw, ih)], &se[0], &se[1], &se[2]);
      for (int c = 0; c < 3; c++) {
        float t = n[c] + w[c] - s[c] - e[c];
        float acc = 0;
        acc += (nw[c] - se[c]) * t;
        t = n[c] - w[c] + e[c] - s[c];
        acc += (ne[c] - sw[c]) * t;
        sum[c] = fmaxf(0.0f, acc);
      }
      float tot = sum[0] + sum[1] + sum[2];
      out[y * iw + x] = tot > 0 ? sqrtf(tot) : 0.0f;
    }
}

/* iu:65-74 */
static inline float cubic1(float p0, float p1, float p2, float p3, float x) {
  float v = p1 - p2, w = p3 - p0;
  float u = v * 3.0f + w;
  u = u * x + (-4.0f * v + (p0 - p1 - w));
  u = u * x + (p2 - p0);
  u = u * x * 0.5f + p1;
  return u;
}

/* iu:87-94 */
static float bicubic(const float *p, float x, float y, int iw, int ih) {
  const int ix = (int)x, iy = (int)y;
  float r[4];
  for (int k = 0; k < 4; k++) {
    int yy = iy - 1 + k;
    r[k] = cubic1(p[mirror2(ix - 1, yy, iw, ih)], p[mirror2(ix, yy, iw, ih)], p[mirror2(ix + 1, yy, iw, ih)], p[mirror2(ix + 2, yy, iw, ih)], x - ix);
  }
  return cubic1(r[0], r[1], r[2], r[3], y - iy);
}

/* iu:456-471 */
void rdo_thinthres(float *out, const float *in, const float *vxy, int iw, int ih) {
  for (int y = 0; y < ih; y++)
    for (int x = 0; x < iw; x++) {
      const int p0 = y * iw + x;
      float vx = vxy[2 * p0], vy = vxy[2 * p0 + 1];
      float am2 = bicubic(in, x - 2 * vx, y - 2 * vy, iw, ih);
      float am1 = bicubic(in, x - 1 * vx, y - 1 * vy, iw, ih);
      float a0 = in[p0];
      float ap1 = bicubic(in, x + 1 * vx, y + 1 * vy, iw, ih);
      float ap2 = bicubic(in, x + 2 * vx, y + 2 * vy, iw, ih);
      out[p0] = (am1 <= a0 && a0 >= ap1) ? (am2 + am1 + a0 + ap1 + ap2) : 0.0f;
    }
}

/* iu:232-237 then iu:211-216 with the literals of rh:262-263 / poly.cpp:114-115 */
void rdo_positive_mask(int *out, const float *in, int n) {
  for (int i = 0; i < n; i++) {
    float t = in[i] > 0.0f ? 1.0f : 0.0f;
    out[i] = (int)(t * 1.0f);
  }
}

/* ------------------------------------------------------------------ connected components */

static int uf_find(int *lab, int a) {
  while (lab[a] != a) a = lab[a];
  return a;
}

static void uf_union(int *lab, int a, int b) {
  a = uf_find(lab, a); b = uf_find(lab, b);
  if (a < b) lab[b] = a; else if (b < a) lab[a] = b;
}

/* iu:495-538 + ih:227-246 run to convergence: 8-connected components of equal pixel value,
 * label = smallest pixel index of the component, pixels equal to bgc get -1.  (The reference does a
 * fixed 10 passes of in-place min-propagation; wherever those converge this is their fixed point.) */
void rdo_label8(int *label, const int *pix, int bgc, int iw, int ih) {
  const int N = iw * ih;
  for (int p = 0; p < N; p++) label[p] = pix[p] == bgc ? -1 : p;
  for (int y = 0; y < ih; y++)
    for (int x = 0; x < iw; x++) {
      const int p = y * iw + x;
      if (label[p] < 0) continue;
      const int v = pix[p];
      if (x > 0 && pix[p - 1] == v) uf_union(label, p, p - 1);
      if (y > 0) {
        if (pix[p - iw] == v) uf_union(label, p, p - iw);
        if (x > 0 && pix[p - iw - 1] == v) uf_union(label, p, p - iw - 1);
        if (x < iw - 1 && pix[p - iw + 1] == v) uf_union(label, p, p - iw + 1);
      }
    }
  for (int p = 0; p < N; p++) if (label[p] >= 0) label[p] = uf_find(label, p);
}

/* iu:641-649.  `out` is accumulated into, exactly like the kernel: the rect path relies on whatever the
 * plane held before, the poly path clears it first (poly.cpp:117). */
void rdo_calc_strength(int *out, const float *edge, const int *label, int iw, int ih) {
  for (int y = 1; y < ih - 1; y++)
    for (int x = 1; x < iw - 1; x++) {
      const int p = y * iw + x;
      if (label[p] <= 0) continue;
      out[label[p]] = (int)((uint32_t)out[label[p]] + (uint32_t)(int)(edge[p] * edge[p] * 10000.0f));
    }
}

/* iu:651-657 */
void rdo_filter_strength(int *label, const int *str, int thre, int iw, int ih) {
  for (int y = 1; y < ih - 1; y++)
    for (int x = 1; x < iw - 1; x++) {
      const int p = y * iw + x;
      if (label[p] <= 0 || str[label[p]] < thre) label[p] = -1;
    }
}

/* iu:225-230 */
void rdo_threshold_i(int *out, const int *in, int lo, int thr, int hi, int n) {
  for (int i = 0; i < n; i++) out[i] = in[i] > thr ? hi : lo;
}

/* ------------------------------------------------------------------ rect-path edge tidy (rc:74-135) */

static const int RX[8] = { 1, 1, 0, -1, -1, -1, 0, 1 }, RY[8] = { 0, -1, -1, -1, 0, 1, 1, 1 };

/* rc:74-95 (positive = `> 0`) and oclpolyline.cl:66-87 (positive = `!= 0`) */
void rdo_junction(int *out, const int *in, int nonzero_variant, int iw, int ih) {
  memset(out, 0, sizeof(int) * (size_t)iw * ih);
  for (int y = 1; y < ih - 1; y++)
    for (int x = 1; x < iw - 1; x++) {
      const int p = y * iw + x;
      const int on = nonzero_variant ? in[p] != 0 : in[p] > 0;
      if (!on) continue;
      int count = 1;
      for (int i = 0; i < 8; i++) {
        int q = in[p + RX[i] + RY[i] * iw];
        if (nonzero_variant ? q != 0 : q > 0) count++;
      }
      out[p] = count == 1 ? 0 : count;
    }
}

/* rc:97-121 */
void rdo_connect_rect(int *out, const int *in, int iw, int ih) {
  memset(out, 0, sizeof(int) * (size_t)iw * ih);
  for (int y = 2; y < ih - 2; y++)
    for (int x = 2; x < iw - 2; x++) {
      const int p = y * iw + x;
      if (in[p] != 0) { out[p] = 1; continue; }
      int o = 0;
      if (in[p - 1] == 2 && in[p + 1] != 0) o = 1;
      if (in[p - 1] != 0 && in[p + 1] == 2) o = 1;
      if (in[p - iw] == 2 && in[p + iw] != 0) o = 1;
      if (in[p - iw] != 0 && in[p + iw] == 2) o = 1;
      if (in[p - iw - 1] == 2 && in[p + iw + 1] == 2) o = 1;
      if (in[p - iw + 1] == 2 && in[p + iw - 1] == 2) o = 1;
      if (in[p + 1] == 2 && in[p + iw - 1] == 2) o = 1;
      if (in[p - 1] == 2 && in[p + iw + 1] == 2) o = 1;
      if (in[p - iw + 1] == 2 && in[p + iw] == 2) o = 1;
      if (in[p - iw - 1] == 2 && in[p + iw] == 2) o = 1;
      out[p] = o;
    }
}

/* rc:123-135 == oclpolyline.cl:112-124 */
void rdo_stringify(int *out, const int *in, int mod2, int iw, int ih) {
  memcpy(out, in, sizeof(int) * (size_t)iw * ih);
  for (int y = 1; y < ih - 1; y++)
    for (int x = 1; x < iw - 1; x++) {
      if (((x + y) & 1) != mod2) continue;
      const int p = y * iw + x;
      const int up = in[p - iw] != 0, dn = in[p + iw] != 0, lf = in[p - 1] != 0, rt = in[p + 1] != 0;
      if ((up && lf) || (up && rt) || (dn && lf) || (dn && rt)) out[p] = 0;
    }
}

/* ------------------------------------------------------------------ edge-stopped box blur (rc:155-205) */

static void blblur_line(uint32_t *out, const int8_t *edge, const uint32_t *in, int iw, int ih, int vertical) {
  const int R = 4;
  for (int y = 0; y < ih; y++)
    for (int x = 0; x < iw; x++) {
      /* c = coordinate along the sweep, o = the other one; E(c) / I(c) read along the sweep */
      const int c0 = vertical ? y : x, n = vertical ? ih : iw, st = vertical ? iw : 1;
      const int base = vertical ? x : y * iw;
      const int p = y * iw + x;
#define E(c) (edge[base + (c) * st])
#define I(c) (in[base + (c) * st])
      /* the "side" neighbour tested in the second stop rule: next line across the sweep */
      const int has_side = vertical ? (x < iw - 1) : (y < ih - 1);
      const int side = vertical ? 1 : iw;
      int wsum = 0, s0 = 0, s1 = 0, s2 = 0;
      const int oe = E(c0) != 0;
      for (int d = 0; d >= -R; d--) {
        const int c = c0 + d;
        if (c < 0) break;
        if (c > 0 && E(c) != 0 && E(c - 1) == 0) break;
        if (c > 0 && has_side && E(c) == 0 && E(c - 1) != 0 && edge[base + c * st + side] != 0) break;
        wsum++;
        uint32_t v = I(c);
        s0 += v & 4095; s1 += (v >> 12) & 1023; s2 += (v >> 22) & 1023;
      }
      for (int d = 0; d <= R; d++) {
        const int c = c0 + d;
        if (c > n - 1) break;
        if (c < n - 1 && E(c) == 0 && E(c + 1) != 0) break;
        if (oe && E(c) == 0) break;
        wsum++;
        uint32_t v = I(c);
        s0 += v & 4095; s1 += (v >> 12) & 1023; s2 += (v >> 22) & 1023;
      }
#undef E
#undef I
      if (wsum == 0) { out[p] = in[p]; continue; }
      uint32_t r = (uint32_t)clampi(s2 / wsum, 0, 1023);
      r = (r << 10) | (uint32_t)clampi(s1 / wsum, 0, 1023);
      r = (r << 12) | (uint32_t)clampi(s0 / wsum, 0, 4095);
      out[p] = r;
    }
}

/* rh:286-296: ten (horizontal, vertical) pairs */
void rdo_blblur(uint32_t *out, const int8_t *edge, const uint32_t *in, int npairs, int iw, int ih) {
  const size_t N = (size_t)iw * ih;
  uint32_t *t = (uint32_t *)malloc(N * 4);
  const uint32_t *src = in;
  for (int i = 0; i < npairs; i++) {
    blblur_line(t, edge, src, iw, ih, 0);
    blblur_line(out, edge, t, iw, ih, 1);
    src = out;
  }
  free(t);
}

/* rc:207-216 */
void rdo_quantize(uint32_t *out, const uint32_t *in, int n0, int n1, int n2, int n) {
  for (int i = 0; i < n; i++) {
    float L, a, b;
    unpack_lab(in[i], &L, &a, &b);
    out[i] = pack_lab(roundf(L * n0) / (float)n0, roundf(a * n1) / (float)n1, roundf(b * n2) / (float)n2);
  }
}

/* rc:218-244 */
void rdo_despeckle(uint32_t *out, const uint32_t *in, const float *edge, int iw, int ih) {
  for (int y = 0; y < ih; y++)
    for (int x = 0; x < iw; x++) {
      const int p0 = y * iw + x;
      out[p0] = in[p0];
      if (edge[p0] < 1e-6f) continue;
      float dist = 1e+10f, l0, a0, b0;
      unpack_lab(in[p0], &l0, &a0, &b0);
      for (int yy = -1; yy <= 1; yy++)
        for (int xx = -1; xx <= 1; xx++) {
          if (x + xx < 0 || x + xx >= iw || y + yy < 0 || y + yy >= ih) continue;
          const int p1 = p0 + yy * iw + xx;
          if (edge[p1] >= 1e-6f) continue;
          float l1, a1, b1;
          unpack_lab(in[p1], &l1, &a1, &b1);
          float dx = l1 - l0, dy = a1 - a0, dz = b1 - b0;
          float d = sqrtf(dx * dx + dy * dy + dz * dz);
          if (d < dist) { out[p0] = in[p1]; dist = d; }
        }
    }
}

/* rc:246-287: set the 4..6 px ring around every pixel with a non-zero junction count, then erase discs
 * (radius 8 around curve ends = count 2, radius 4 around the others).  `out` must be zero on entry. */
void rdo_merge_mask(int *out, const int *junction, int iw, int ih) {
  for (int pass = 0; pass < 2; pass++)
    for (int y = 0; y < ih; y++)
      for (int x = 0; x < iw; x++) {
        const int j = junction[y * iw + x];
        if (j == 0) continue;
        const int r = pass == 0 ? 6 : (j == 2 ? 8 : 4);
        for (int yy = y - r; yy <= y + r; yy++)
          for (int xx = x - r; xx <= x + r; xx++) {
            if (xx < 0 || xx >= iw || yy < 0 || yy >= ih) continue;
            const int d2 = (yy - y) * (yy - y) + (xx - x) * (xx - x);
            if (pass == 0) { if (16 <= d2 && d2 < 36) out[yy * iw + xx] = 1; }
            else if (d2 < (j == 2 ? 64 : 16)) out[yy * iw + xx] = 0;
          }
      }
}

/* rc:289-298 */
void rdo_region_label_init(int *label, const int *pix, int iw, int ih) {
  for (int y = 0; y < ih; y++)
    for (int x = 0; x < iw; x++) {
      const int p = y * iw + x;
      if (y > 0 && pix[p] == pix[p - iw]) label[p] = p - iw;
      else if (x > 0 && pix[p] == pix[p - 1]) label[p] = p - 1;
      else label[p] = p;
    }
}

/* rc:300-334, one in-place pass in raster order (canonical order) */
void rdo_region_merge_pass(int *label, const int *pix, const int *mask, const int *edge, int iw, int ih) {
  for (int y = 1; y < ih - 1; y++)
    for (int x = 1; x < iw - 1; x++) {
      const int p0 = y * iw + x;
      int g = label[p0];
      const int og = g;
      if (g == -1) continue;
      const int any = mask[p0] != 0;
      int p1, s;
      p1 = p0 - iw; s = label[p1];
      if (s < g && (pix[p0] == pix[p1] || any) && edge[p0] <= 0) g = s;
      p1 = p0 - 1; s = label[p1];
      if (s < g && (pix[p0] == pix[p1] || any) && edge[p0] <= 0) g = s;
      p1 = p0 + 1; s = label[p1];
      if (s < g && (pix[p0] == pix[p1] || any) && edge[p1] <= 0) g = s;
      p1 = p0 + iw; s = label[p1];
      if (s < g && (pix[p0] == pix[p1] || any) && edge[p1] <= 0) g = s;
      for (int j = 0; j < 8; j++) g = label[g];
      if (g != og) {
        if (g < label[og]) label[og] = g;
        if (g < label[p0]) label[p0] = g;
      }
    }
}

/* rc:336-346.  Accumulates into `out` (the rect path does not clear it). */
void rdo_region_size(int *out, const int *label, int n) {
  for (int i = 0; i < n; i++) if (label[i] != -1) out[label[i]]++;
}

/* rc:348-371, in place in raster order */
void rdo_despeckle2(int *label, const int *size, int thre, int iw, int ih) {
  for (int y = 0; y < ih; y++)
    for (int x = 0; x < iw; x++) {
      const int p0 = y * iw + x;
      if (size[label[p0]] > thre) continue;
      int maxSize = 0, maxLabel = label[p0];
      for (int yy = -1; yy <= 1; yy++)
        for (int xx = -1; xx <= 1; xx++) {
          if (x + xx < 0 || x + xx >= iw || y + yy < 0 || y + yy >= ih) continue;
          const int l1 = label[p0 + yy * iw + xx];
          if (size[l1] > maxSize) { maxSize = size[l1]; maxLabel = l1; }
        }
      label[p0] = maxLabel;
    }
}

/* rc:373-390 */
void rdo_mark_boundary(int *out, const int *in, int iw, int ih) {
  for (int y = 0; y < ih; y++)
    for (int x = 0; x < iw; x++) {
      const int p0 = y * iw + x;
      if (x <= 1 || y <= 1 || x >= iw - 2 || y >= ih - 2) { out[p0] = -1; continue; }
      int near = 0;
      for (int yy = -2; yy <= 2 && !near; yy++)
        for (int xx = -2; xx <= 2; xx++)
          if (in[p0 + yy * iw + xx] != in[p0]) { near = 1; break; }
      out[p0] = near ? in[p0] : -1;
    }
}

/* rc:426-464 in raster order.  table: nentry * 5 ints, zero on entry. */
void rdo_reduce_ls(int *table, const int *boundary, const int *lsid, int iw, int ih, int nentry) {
  for (int y = 1; y < ih - 1; y++)
    for (int x = 1; x < iw - 1; x++) {
      const int id = lsid[y * iw + x];
      if (id <= 0) continue;
      int bid = 0;
      unsigned slot = 0;
      for (int yy = -3; yy <= 3; yy++) {
        if (y + yy < 0 || y + yy >= ih) continue;
        for (int xx = -3; xx <= 3; xx++) {
          if (x + xx < 0 || x + xx >= iw) continue;
          const int b = boundary[(y + yy) * iw + x + xx];
          if (b <= 0) continue;
          if (bid != b) {
            bid = b;
            slot = (((unsigned)id * (unsigned)bid) & 0x7fffffffu) % (unsigned)nentry;
          }
          int *e = table + (size_t)slot * 5;
          if (e[0] == 0) { e[0] = id; continue; }  /* the claiming visit does not update the box */
          if (e[0] != id) continue;
          if (iw - x > e[1]) e[1] = iw - x;
          if (x > e[2]) e[2] = x;
          if (ih - y > e[3]) e[3] = ih - y;
          if (y > e[4]) e[4] = y;
        }
      }
    }
}

/* ------------------------------------------------------------------ whole rect-path device stage (rh:235-381) */

static void *zalloc(size_t n) { return calloc(1, n ? n : 1); }

rdo_rect_t *rdo_rect_new(int iw, int ih) {
  rdo_rect_t *c = (rdo_rect_t *)zalloc(sizeof(*c));
  const size_t N = (size_t)iw * ih;
  c->iw = iw; c->ih = ih;
  c->plab0 = zalloc(N * 4); c->plab1 = zalloc(N * 4); c->Lblur = zalloc(N * 4);
  c->vxy = zalloc(N * 8); c->strength = zalloc(N * 4); c->nms = zalloc(N * 4);
  c->mask0 = zalloc(N * 4); c->tidy = zalloc(N * 4); c->label1 = zalloc(N * 4);
  c->str_sum = zalloc(N * 4); c->prev_strong = zalloc(N * 4); c->edge500 = zalloc(N * 4);
  c->smooth = zalloc(N * 4); c->quant = zalloc(N * 4); c->strong = zalloc(N * 4);
  c->junction = zalloc(N * 4); c->mergemask = zalloc(N * 4); c->region = zalloc(N * 4);
  c->rsize = zalloc(N * 4); c->boundary_src = zalloc(N * 4); c->boundary = zalloc(N * 4);
  c->lsid = zalloc(N * 4); c->lslist = zalloc(N * 16); c->table = zalloc(N * 16);
  return c;
}

void rdo_rect_free(rdo_rect_t *c) {
  void **p = (void **)&c->plab0;
  (void)p;
  free(c->plab0); free(c->plab1); free(c->Lblur); free(c->vxy); free(c->strength); free(c->nms);
  free(c->mask0); free(c->tidy); free(c->label1); free(c->str_sum); free(c->prev_strong); free(c->edge500);
  free(c->smooth); free(c->quant); free(c->strong); free(c->junction); free(c->mergemask); free(c->region);
  free(c->rsize); free(c->boundary_src); free(c->boundary); free(c->lsid); free(c->lslist); free(c->table);
  free(c);
}

void *rdo_rect_plane(rdo_rect_t *c, const char *name) {
#define P(n) if (!strcmp(name, #n)) return c->n;
  P(plab0) P(plab1) P(Lblur) P(vxy) P(strength) P(nms) P(mask0) P(tidy) P(label1) P(str_sum) P(prev_strong)
  P(edge500) P(smooth) P(quant) P(strong) P(junction) P(mergemask) P(region) P(rsize) P(boundary_src)
  P(boundary) P(lsid) P(lslist) P(table)
#undef P
  return NULL;
}

void rdo_rect_set_region_mode(rdo_rect_t *c, int mode) { c->region_mode = mode; }
int rdo_rect_info(const rdo_rect_t *c, int which) { return which == 0 ? c->region_mode : which == 1 ? c->region_rounds : which == 2 ? c->absorb_rounds : -1; }

void rdo_rect_frame(rdo_rect_t *c, const uint8_t *bgr, int ws) {
  const int iw = c->iw, ih = c->ih, N = iw * ih;
  float *fa = (float *)malloc(sizeof(float) * N), *fb = (float *)malloc(sizeof(float) * N);
  float *ba = (float *)malloc(sizeof(float) * N), *bb = (float *)malloc(sizeof(float) * N);
  int *t0 = (int *)malloc(sizeof(int) * N), *t1 = (int *)malloc(sizeof(int) * N);
  int8_t *e8 = (int8_t *)malloc(N);

  /* rh:245-258 colour, blur, gradient, non-max suppression */
  rdo_bgr2plab(c->plab0, bgr, iw, ih, ws);
  rdo_unpack_plab(c->Lblur, fa, fb, c->plab0, N);         /* Lblur used as scratch for the raw L */
  rdo_iirblur(bb, fb, iw, ih);
  rdo_iirblur(ba, fa, iw, ih);
  memcpy(fa, c->Lblur, sizeof(float) * N);
  rdo_iirblur(c->Lblur, fa, iw, ih);
  rdo_pack_plab(c->plab1, c->Lblur, ba, bb, N);
  rdo_edgevec(c->vxy, c->Lblur, iw, ih);
  rdo_edge_plab(c->strength, c->plab1, iw, ih);
  rdo_thinthres(c->nms, c->strength, c->vxy, iw, ih);

  /* rh:262-272 mask and looser tidy */
  rdo_positive_mask(c->mask0, c->nms, N);
  rdo_junction(t0, c->mask0, 0, iw, ih);
  rdo_connect_rect(t1, t0, iw, ih);
  rdo_stringify(t0, t1, 0, iw, ih);
  rdo_stringify(c->tidy, t0, 1, iw, ih);

  /* rh:274-284 components of the tidied mask (background labelled too), strength filter at 500 */
  rdo_label8(c->label1, c->tidy, -1, iw, ih);
  memcpy(c->str_sum, c->prev_strong, sizeof(int) * N);     /* H1: plane still holds last frame's strong mask */
  rdo_calc_strength(c->str_sum, c->nms, c->label1, iw, ih);
  rdo_filter_strength(c->label1, c->str_sum, 500, iw, ih);
  rdo_threshold_i(c->edge500, c->label1, 0, 0, 1, N);
  for (int i = 0; i < N; i++) e8[i] = (int8_t)c->edge500[i];

  /* rh:286-303 edge-preserving smoothing, quantisation, despeckle */
  rdo_blblur(c->smooth, e8, c->plab0, 10, iw, ih);
  rdo_quantize((uint32_t *)t0, c->smooth, 24, 24, 24, N);
  rdo_despeckle(c->quant, (uint32_t *)t0, c->nms, iw, ih);

  /* rh:307-321 strong edges, junction counts, merge mask */
  rdo_filter_strength(c->label1, c->str_sum, 2500, iw, ih);
  rdo_threshold_i(c->strong, c->label1, 0, 0, 1, N);
  memcpy(c->prev_strong, c->strong, sizeof(int) * N);
  rdo_junction(c->junction, c->label1, 0, iw, ih);
  memset(c->mergemask, 0, sizeof(int) * N);
  rdo_merge_mask(c->mergemask, c->junction, iw, ih);

  /* rh:325-336 regions */
  if (c->region_mode == 0) {                               /* the reference in serial raster order */
    rdo_region_label_init(c->region, (const int *)c->quant, iw, ih);
    for (int i = 0; i < 8; i++) rdo_region_merge_pass(c->region, (const int *)c->quant, c->mergemask, c->label1, iw, ih);
    c->region_rounds = 8;
  } else                                                   /* the reference with concurrent work-items: 2 = its 8 launches, 1 = launched until nothing changes (SPEC) */
    c->region_rounds = rdo_region_concurrent(c->region, (const int *)c->quant, c->mergemask, c->label1, iw, ih, c->region_mode == 2 ? 8 : RDO_REGION_MAX_LAUNCHES);
  memcpy(c->rsize, c->junction, sizeof(int) * N);          /* H2: size plane still holds the junction counts */
  rdo_region_size(c->rsize, c->region, N);
  rdo_despeckle2(c->region, c->rsize, 16, iw, ih);        /* both modes: the serial raster order's result (what the HIP path computes exactly) */
  c->absorb_rounds = 0;

  /* rh:340-342 region boundaries and their components */
  rdo_mark_boundary(c->boundary_src, c->region, iw, ih);
  rdo_label8(c->boundary, c->boundary_src, -1, iw, ih);

  /* rh:361 polylines on the strong mask (ring of the stale scratch plane is non-zero, H3) */
  rdo_polyline(c->lslist, N * 16, c->lsid, c->strong, 1, 4.0f, 20, iw, ih, NULL);

  /* rh:365-367 */
  memset(c->table, 0, (size_t)N * 16);
  rdo_reduce_ls(c->table, c->boundary, c->lsid, iw, ih, N * 4 / 5);

  free(fa); free(fb); free(ba); free(bb); free(t0); free(t1); free(e8);
}

/* ------------------------------------------------------------------ SPEC of the region stages as the HIP path evaluates them
 *
 * The reference's labelMergeMain (rc:300-334, 8 in-place launches) gives a result that depends on the order in which a device
 * runs the work-items: the serial raster order of rdo_region_merge_pass above is one legal order, and no parallel
 * schedule reproduces it.  The HIP path reproduces ANOTHER legal execution of the same kernel - all work-items of a launch
 * concurrent - bit for bit, and keeps launching until a launch changes nothing ("region mode 1" of rdo_rect_frame; mode 2
 * stops after the reference's 8 launches and is what oracle/_ref computes under rdcl_set_order(..., 0, 0, 5, 0)).
 * despeckle2 (rc:348-371, in place; H6) is order dependent as well, but the serial raster order's result is a recurrence the
 * HIP path evaluates EXACTLY (rd_k_rect.hip: k_absorb_tile / k_absorb_tail), so every mode uses rdo_despeckle2.
 * GPU tests compare the region, rsize, boundary_src, boundary and table planes bit for bit against mode 1. */

/* rc:289-334 with the work-items of every launch running CONCURRENTLY: all of them read the labels the launch began with, and
 * their atomic minima take effect together when it ends - a legal execution of the reference's kernel on a device (OpenCL gives
 * no guarantee that a work-item sees another one's update within a launch), and the one a parallel device can reproduce bit for
 * bit.  `launches` = 8 is what the reference enqueues (rh:325-331) - enough in serial orders, where a launch carries a label
 * across the frame; concurrent work-items pass it on one hop (plus pointer jumps) per launch, and a 1920x1080 frame settles
 * after about 12 - so the SPEC keeps launching until a launch changes nothing (every later launch would repeat it), at most
 * `launches`.  The labels start from the raw links of rc:289-298 (the labels a work-item compares are whatever the trees hold at
 * that moment, not their roots: which regions the merge mask joins depends on that), and the 8 pointer jumps of a work-item
 * follow the labels as they were when the launch began.  Pixels on the frame's ring are never processed (rc:302) and keep
 * their links unless a work-item hooks them.  Returns the number of launches evaluated.
 * oracle/_ref runs the reference's own kernel this way under rdcl_set_order(..., 0, 0, 5, 0); tests/test_cpu_oracle.py
 * compares the planes of 8 launches. */
int rdo_region_concurrent(int *label, const int *pix, const int *mask, const int *edge, int iw, int ih, int launches) {
  const int N = iw * ih;
  rdo_region_label_init(label, pix, iw, ih);
  int *nxt = (int *)malloc(sizeof(int) * N);
  int l = 0;
  for (; l < launches; l++) {
    int changed = 0;
    memcpy(nxt, label, sizeof(int) * N);
    for (int y = 1; y < ih - 1; y++)
      for (int x = 1; x < iw - 1; x++) {
        const int p0 = y * iw + x;
        const int og = label[p0];
        if (og == -1) continue;
        int g = og;
        const int any = mask[p0] != 0;
        int p1, s;
        p1 = p0 - iw; s = label[p1]; if (s < g && (pix[p0] == pix[p1] || any) && edge[p0] <= 0) g = s;
        p1 = p0 - 1;  s = label[p1]; if (s < g && (pix[p0] == pix[p1] || any) && edge[p0] <= 0) g = s;
        p1 = p0 + 1;  s = label[p1]; if (s < g && (pix[p0] == pix[p1] || any) && edge[p1] <= 0) g = s;
        p1 = p0 + iw; s = label[p1]; if (s < g && (pix[p0] == pix[p1] || any) && edge[p1] <= 0) g = s;
        for (int j = 0; j < 8; j++) g = label[g];
        if (g != og) {
          if (g < nxt[og]) nxt[og] = g;
          if (g < nxt[p0]) nxt[p0] = g;
          changed = 1;
        }
      }
    memcpy(label, nxt, sizeof(int) * N);
    if (!changed) { l++; break; }       /* (a launch that changes nothing: every later one would repeat it) */
  }
  free(nxt);
  return l;
}

/* rc:348-371 as `max_rounds` JACOBI rounds of the raster-order recurrence: in round r+1 a small-region pixel picks the
 * largest region among its 3x3 neighbourhood, reading the round-r labels of the four neighbours that precede it in raster
 * order (NW, N, NE, W) and the input labels of the others.  The fixed point of this iteration IS the serial raster result
 * (rdo_despeckle2; CPU test) - that is the argument by which the HIP path's parallel evaluation is exact.  Returns the rounds
 * evaluated, *nsmall = pixels of small regions. */
int rdo_despeckle2_jacobi_k(int *label, const int *size, int thre, int iw, int ih, int *nsmall, int max_rounds) {
  const int N = iw * ih;
  int *old = (int *)malloc(sizeof(int) * N), *cur = (int *)malloc(sizeof(int) * N), *nxt = (int *)malloc(sizeof(int) * N);
  memcpy(old, label, sizeof(int) * N); memcpy(cur, label, sizeof(int) * N);
  int rounds = 0, ns = 0;
  for (int p = 0; p < N; p++) if (size[old[p]] <= thre) ns++;
  if (nsmall) *nsmall = ns;
  while (rounds < max_rounds) {
    int changed = 0;
    memcpy(nxt, cur, sizeof(int) * N);
    for (int y = 0; y < ih; y++)
      for (int x = 0; x < iw; x++) {
        const int p0 = y * iw + x;
        if (size[old[p0]] > thre) continue;
        int maxSize = 0, maxLabel = old[p0];
        for (int yy = -1; yy <= 1; yy++)
          for (int xx = -1; xx <= 1; xx++) {
            if (x + xx < 0 || x + xx >= iw || y + yy < 0 || y + yy >= ih) continue;
            const int p1 = p0 + yy * iw + xx;
            const int l1 = (yy < 0 || (yy == 0 && xx < 0)) ? cur[p1] : old[p1];
            if (size[l1] > maxSize) { maxSize = size[l1]; maxLabel = l1; }
          }
        if (nxt[p0] != maxLabel) { nxt[p0] = maxLabel; changed++; }
      }
    memcpy(cur, nxt, sizeof(int) * N);
    rounds++;
    if (!changed) break;
  }
  memcpy(label, cur, sizeof(int) * N);
  free(old); free(cur); free(nxt);
  return rounds;
}

/* experiment: chunked evaluation of one IIR sweep.  Chunk c covers logical steps [c*C, (c+1)*C); it starts Wm steps
 * earlier with zero state (real inputs) and its results are accepted as-is.  Returns the number of output samples
 * that differ (bitwise) from the full sweep; *unverified counts chunk starts whose 7 preceding outputs differ. */
int rdo_iir_chunk_test(const float *src, int n, int st, int dir, int C, int Wm, int *unverified) {
  float *full = (float *)malloc(sizeof(float) * (size_t)n * st + 64), *chk = (float *)malloc(sizeof(float) * (size_t)n * st + 64);
  memset(full, 0, sizeof(float) * (size_t)n * st);
  memset(chk, 0, sizeof(float) * (size_t)n * st);
  iir_sweep(full, src, n, st, dir, IIR, IIR_WARM);
  const int count = n + IIR_WARM + (dir > 0 ? 0 : 1);
  const int x0 = dir > 0 ? -IIR_WARM : n + IIR_WARM;
  int bad = 0, unv = 0;
  for (int c0 = 0; c0 < count; c0 += C) {
    const int begin = c0 == 0 ? 0 : (c0 - Wm < 0 ? 0 : c0 - Wm);
    float iv[8] = { 0 }, tv[8] = { 0 };
    float tail_w[7]; int ok = 1;
    for (int s = begin; s < count && s < c0 + C; s++) {
      const int x = x0 + s * dir;
      iv[0] = src[mirror1(x, n) * st];
      float d = iv[0] * IIR[0];
      d += IIR[1] * iv[1] + IIR[2] * iv[2] + IIR[3] * iv[3] + IIR[4] * iv[4] + IIR[5] * iv[5] + IIR[6] * iv[6] + IIR[7] * iv[7];
      d += IIR[8] * tv[0] + IIR[9] * tv[1] + IIR[10] * tv[2] + IIR[11] * tv[3] + IIR[12] * tv[4] + IIR[13] * tv[5] + IIR[14] * tv[6];
      if (s >= c0 && x >= 0 && x < n) chk[x * st] = d;
      if (s < c0 && s >= c0 - 7) tail_w[s - (c0 - 7)] = d;
      for (int k = 7; k > 0; k--) { iv[k] = iv[k - 1]; tv[k] = tv[k - 1]; }
      tv[0] = d;
    }
    (void)tail_w; (void)ok;
  }
  for (int x = 0; x < n; x++) if (memcmp(&full[x * st], &chk[x * st], 4) != 0) bad++;
  /* a chunk start is "verified" when the 7 outputs before it agree between the warm-up run and the true sweep; here we
   * simply count chunks containing a wrong sample */
  for (int c0 = C; c0 < count; c0 += C) {
    int wrong = 0;
    for (int s = c0; s < count && s < c0 + C; s++) { const int x = x0 + s * dir; if (x >= 0 && x < n && memcmp(&full[x * st], &chk[x * st], 4) != 0) wrong = 1; }
    unv += wrong;
  }
  *unverified = unv;
  free(full); free(chk);
  return bad;
}


/* ================================================================================================ visualisers and the
 * operators no application of the reference calls (oclimgutil.h:84-98).  Restated for completeness of the oclimgutil.h
 * surface; pinned against the reference's own kernels by tests/golden/ops_*.npz (tools/make_golden_ops.py). */

/* iu:283-289 */
void rdo_convert_bgr_lumaf(uint8_t *out, const float *in, float f, int iw, int ih, int ws) {
  for (int y = 0; y < ih; y++)
    for (int x = 0; x < iw; x++) {
      const int v = clampi((int)floorf(in[y * iw + x] * f * 255), 0, 255);
      uint8_t *o = out + (size_t)y * ws + x * 3;
      o[0] = o[1] = o[2] = (uint8_t)v;
    }
}

/* iu:291-321 */
void rdo_convert_bgr_labeli(uint8_t *out, const int *in, int bgc, int iw, int ih, int ws) {
  for (int y = 0; y < ih; y++)
    for (int x = 0; x < iw; x++) {
      const int c = in[y * iw + x];
      uint8_t *o = out + (size_t)y * ws + x * 3;
      if (c == bgc) { o[0] = o[1] = o[2] = 0; continue; }
      const int g = (int)((unsigned)c * 1103515245u + 12345u);
      o[2] = (uint8_t)((((g & (7 << 0)) << 5) | 31) & 255);
      o[1] = (uint8_t)((((g & (7 << 3)) << 2) | 31) & 255);
      o[0] = (uint8_t)((((g & (7 << 6)) >> 1) | 31) & 255);
    }
}

/* iu:136-182: packed Lab -> sRGB bytes */
static inline float icfunc(float ft) { return ft > 0.20689270648f ? ft * ft * ft : (ft - 16.0f / 116) * (1.0f / 7.787f); }

void rdo_plab2bgr(uint8_t *out, const uint32_t *in, int iw, int ih, int ws) {
  init_luts();
  const float xn = 0.950456f, zn = 1.088754f;
  for (int y = 0; y < ih; y++)
    for (int x = 0; x < iw; x++) {
      float L, a, b;
      unpack_lab(in[y * iw + x], &L, &a, &b);
      L *= 256; a *= 256; b *= 256;
      float cy;
      if (L > 0.20689270648f) { cy = (L + 16) * (1.0f / 116.0f); cy = cy * cy * cy; }
      else cy = L * (1.0f / 903.3f);
      const float fy = (float)(lut_cf[clampi((int)floorf(cy * 1024), 0, 1023)] + 9039) * (1.0f / 65536.0f);
      const float fz = fy - (b - 128) * (1.0f / 200.0f);
      const float fx = fy + (a - 128) * (1.0f / 500.0f);
      const float cx = icfunc(fx) * xn, cz = icfunc(fz) * zn;
      const float r = cx * 3.240479f + cy * -1.537150f + cz * -0.498535f;
      const float g = cx * -0.969256f + cy * 1.875991f + cz * 0.041556f;
      const float bb = cx * 0.055648f + cy * -0.204043f + cz * 1.057311f;
      uint8_t *o = out + (size_t)y * ws + x * 3;
      o[2] = (uint8_t)lut_l2s[clampi((int)floorf(r * 1024), 0, 1023)];
      o[1] = (uint8_t)lut_l2s[clampi((int)floorf(g * 1024), 0, 1023)];
      o[0] = (uint8_t)lut_l2s[clampi((int)floorf(bb * 1024), 0, 1023)];
    }
}

/* iu:439-453 */
void rdo_edge_f_f(float *out, const float *in, int iw, int ih) {
  for (int y = 0; y < ih; y++)
    for (int x = 0; x < iw; x++) {
      float sum = 0, t;
      t = in[mirror2(x, y - 1, iw, ih)] + in[mirror2(x - 1, y, iw, ih)] - in[mirror2(x, y + 1, iw, ih)] - in[mirror2(x + 1, y, iw, ih)];
      sum += (in[mirror2(x - 1, y - 1, iw, ih)] - in[mirror2(x + 1, y + 1, iw, ih)]) * t;
      t = in[mirror2(x, y - 1, iw, ih)] - in[mirror2(x - 1, y, iw, ih)] + in[mirror2(x + 1, y, iw, ih)] - in[mirror2(x, y + 1, iw, ih)];
      sum += (in[mirror2(x + 1, y - 1, iw, ih)] - in[mirror2(x - 1, y + 1, iw, ih)]) * t;
      out[y * iw + x] = sqrtf(fmaxf(0.0f, sum));
    }
}

/* iu:354-393: gradient direction of the channel with the largest response, sign taken from the L channel */
void rdo_edgevec_plab(float *vxy, const uint32_t *in, int iw, int ih) {
  for (int y = 0; y < ih; y++)
    for (int x = 0; x < iw; x++) {
      float vx3[3] = { 0, 0, 0 }, vy3[3] = { 0, 0, 0 };
      for (int yy = -2; yy <= 2; yy++)
        for (int xx = -2; xx <= 2; xx++) {
          float s[3];
          unpack_lab(in[mirror2(x + xx, y + yy, iw, ih)], &s[0], &s[1], &s[2]);
          for (int c = 0; c < 3; c++) {
            vx3[c] += V5C[(xx + 2) + (yy + 2) * 5] * s[c];
            vy3[c] += V5C[(yy + 2) + (xx + 2) * 5] * s[c];
          }
        }
      float l3[3];
      for (int c = 0; c < 3; c++) l3[c] = vx3[c] * vx3[c] + vy3[c] * vy3[c];
      float ivlen, vx, vy;
      if (l3[0] >= l3[1] && l3[0] >= l3[2]) { ivlen = l3[0]; vx = vx3[0]; vy = vy3[0]; }
      else if (l3[1] >= l3[2]) { ivlen = l3[1]; vx = vx3[1]; vy = vy3[1]; }
      else { ivlen = l3[2]; vx = vx3[2]; vy = vy3[2]; }
      if ((double)l3[0] >= 1e-6 && (vx3[0] * vx + vy3[0] * vy < 0)) { vx = -vx; vy = -vy; }
      if ((double)ivlen > 1e-10) { ivlen = 1.0f / sqrtf(ivlen); vx *= ivlen; vy *= ivlen; }
      else vx = vy = 0.70710678118f;
      vxy[(y * iw + x) * 2] = vx; vxy[(y * iw + x) * 2 + 1] = vy;
    }
}

/* iu:473-491: like thinthres, with a 1 % tolerance and all four samples in the comparison */
void rdo_thincubic(float *out, const float *in, const float *vxy, int iw, int ih) {
  for (int y = 0; y < ih; y++)
    for (int x = 0; x < iw; x++) {
      const int p0 = y * iw + x;
      const float vx = vxy[p0 * 2], vy = vxy[p0 * 2 + 1];
      const float am2 = bicubic(in, x - 2 * vx, y - 2 * vy, iw, ih);
      const float am1 = bicubic(in, x - 1 * vx, y - 1 * vy, iw, ih);
      const float a0 = in[p0];
      const float ap1 = bicubic(in, x + 1 * vx, y + 1 * vy, iw, ih);
      const float ap2 = bicubic(in, x + 2 * vx, y + 2 * vy, iw, ih);
      const float C = 0.99f;
      out[p0] = (am2 * C <= a0 && am1 * C <= a0 && a0 >= ap1 * C && a0 >= ap2 * C) ? (am2 + am1 + a0 + ap1 + ap2) : 0;
    }
}


/* oclrect.c:1066-1083 (see rd_oracle.h).  vec234.h:37-42: a vector is normalised by multiplying with 1 / (sqrt(sum of squares) + 1e-20). */
void rdo_probe_pixels(float fx0, float fy0, float fx1, float fy1, int iw, int ih, int *out) {
  const double ax = rint((double)fx0), ay = rint((double)fy0), bx = rint((double)fx1), by = rint((double)fy1);
  const double ex = bx - ax, ey = by - ay;
  const double inv = 1.0 / (sqrt(ex * ex + ey * ey) + 1e-20);
  const double dx = ex * inv, dy = ey * inv;
  const double nx = -dy, ny = dx;                        /* the segment's normal */
  for (int j = 0; j < 3; j++) {
    const double t = (j + 0.5) / 3;
    const double px = ax + ex * t, py = ay + ey * t;
    for (int dist = -2; dist <= 2; dist++) {
      const double cx = px + nx * dist, cy = py + ny * dist;
      const int x = (int)(cx + 0.5), y = (int)(cy + 0.5);
      int *o = out + 2 * (j * 5 + dist + 2);
      if (x < 0 || x >= iw || y < 0 || y >= ih) { o[0] = -1; o[1] = -1; }
      else { o[0] = x; o[1] = y; }
    }
  }
}
