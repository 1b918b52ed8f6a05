// rectdetect-mi355x: gradient direction + re-packed blurred Lab + edge strength + non-max suppression of the frame path as ONE tile kernel (gfx950).
// Reference behaviour being reproduced: oclimgutil.cl ("iu") 395-471.  Built with -fno-slp-vectorize (see rd_front_helpers.h, Makefile).
#include "rd_device.h"
#include "rd_kernels.h"
#include "rd_front_helpers.h"

namespace {

using namespace rd;

struct P3c { const float *p[3]; };
inline int cdiv(int a, int b) { return (a + b - 1) / b; }
const dim3 block2(64, 4);

// ------------------------------------------------------------------------------------------------ gradient + strength + non-max suppression, one tile kernel
// The frame path's form of iu:395-471 (edgevec_f, the packing of the blurred Lab, edge_plab, thinthres_f_f_f2): ONE launch that reads the three
// blurred planes once and writes the suppressed strength - the direction (8 B/pixel), the re-packed blurred Lab and the strength plane
// (4 B/pixel each) never leave the chip.  A block owns 64 x GN_ROWS pixels.  What a pixel's result depends on:
//   suppression : strength within [-3, +4] of the pixel in x and y (two bicubic footprints a step along the direction, two more two steps away)
//   strength    : the re-packed Lab of the 8 neighbours                       -> Lab within [-4, +5]
//   direction   : the blurred L (raw float) within +-2
// so the block stages a (64 + 9) x (GN_ROWS + 9) tile of L, a, b (mirrored addresses, iu:41-49), keeps the raw L and the three fields as the
// floats unpack_lab(pack_lab()) makes of them, evaluates the strength for the (64 + 7) x (GN_ROWS + 7) cells the suppression reads and then
// suppresses from LDS exactly like k_thinthres.  A strength cell outside the frame is the strength AT ITS MIRRORED POSITION (what
// k_thinthres' staging reads), not the formula applied to mirrored neighbours: border tiles evaluate such a cell around the mirrored centre.
// Same operations in the same order per value as the three kernels; TAPS = 1 also writes the three intermediate planes (tests, debug planes).
#ifndef GN_ROWS
#define GN_ROWS 16
#endif
#define GN_LW 73                       // Lab tile: columns x0 - 4 .. x0 + 68
#define GN_LH (GN_ROWS + 9)
#define GN_SW 71                       // strength tile: columns x0 - 3 .. x0 + 67 (pitch TT_PITCH)
#define GN_SH (GN_ROWS + 7)
#ifndef GN_TY
#define GN_TY 4                       // thread rows (waves) per block
#endif
#define GN_NT (64 * GN_TY)
#define GN_K (GN_ROWS / GN_TY)         // pixels per thread: a column of GN_K rows
struct GnTaps { uint32_t *plab1; float2 *vxy; float *strength; };

template <int TAPS>
__global__ __launch_bounds__(GN_NT) void k_grad_nms(float *__restrict__ out, P3c bl, int iw, int ih, size_t zs, GnTaps taps, int gdim) {
  const rd_tile rd_b = rd_block_tile(gdim);
  if (rd_b.x < 0) return;
  RD_ZSHIFTZ(rd_b.z, zs, out, bl.p[0], bl.p[1], bl.p[2], taps.plab1, taps.vxy, taps.strength);
  // [0] raw L, [1..3] the re-packed fields as floats; after the strength is known the same memory holds the list of local maxima
  __shared__ float lab[4][GN_LH * GN_LW];
  __shared__ float str[GN_SH * TT_PITCH];
  __shared__ int nlst;
  static_assert(4 * GN_LH * GN_LW * 4 >= 64 * GN_ROWS * 20, "the list of maxima lives in the Lab tile's memory");
  const int x0 = rd_b.x * 64, y0 = rd_b.y * GN_ROWS;
  const int tx = threadIdx.x, ty = rd_ty(), tid = ty * 64 + tx;
  if (tid == 0) nlst = 0;
  // ---- stage L, a, b (+ halo): a wave takes every fourth row of the tile, its lanes columns 0..63 (row address uniform, column address fixed per
  // lane); the 9 columns left over are a cell per thread.  All loads of a thread are in flight before the first is used.
  {
    constexpr int NR = (GN_LH + GN_TY - 1) / GN_TY;        // rows per wave
    const int mcol = mirror1(x0 - 4 + tx, iw);                                   // (x0 - 4 + tx <= iw + 4 for every lane that matters: see below)
    const int sc = tid < 9 * GN_LH ? tid : 0, scy = sc / 9, scx = 64 + sc - scy * 9;      // the thread's cell of the side strip
    float vl[NR + 1], va[NR + 1], vb[NR + 1];
#pragma unroll
    for (int i = 0; i <= NR; i++) {
      const int cy = i < NR ? ty + GN_TY * i : scy, cx = i < NR ? tx : scx;
      const int fx = x0 - 4 + cx, fy = y0 - 4 + cy;
      const bool ok = cy < GN_LH && fx <= iw + 4 && fy <= ih + 4;      // (cells further out are read by no pixel of the frame)
      const int a = ok ? mirror1(fy, ih) * iw + (i < NR ? mcol : mirror1(fx, iw)) : 0;
      vl[i] = bl.p[0][a]; va[i] = bl.p[1][a]; vb[i] = bl.p[2][a];
    }
#pragma unroll
    for (int i = 0; i <= NR; i++) {
      const int cy = i < NR ? ty + GN_TY * i : scy, cx = i < NR ? tx : scx;
      if (cy >= GN_LH || (i == NR && tid >= 9 * GN_LH)) continue;
      // pack_lab + unpack_lab (iu:28-39) with the integer fields kept: floor, clamp (NaN and negatives -> 0, large -> the field's maximum), then
      // field * 2^-k + half an LSB
      const int il = clampi(__float2int_rd(vl[i] * 4096), 0, 4095), ia = clampi(__float2int_rd(va[i] * 1024), 0, 1023), ib = clampi(__float2int_rd(vb[i] * 1024), 0, 1023);
      const int t = cy * GN_LW + cx;
      lab[0][t] = vl[i];
      lab[1][t] = (float)il * (1.0f / 4096) + 0.5f / 4096; lab[2][t] = (float)ia * (1.0f / 1024) + 0.5f / 1024; lab[3][t] = (float)ib * (1.0f / 1024) + 0.5f / 1024;
      if (TAPS) {
        const int fx = x0 - 4 + cx, fy = y0 - 4 + cy;
        if (cx >= 4 && cx < 68 && cy >= 4 && cy < 4 + GN_ROWS && fx < iw && fy < ih) taps.plab1[fy * iw + fx] = (uint32_t)il | ((uint32_t)ia << 12) | ((uint32_t)ib << 22);
      }
    }
  }
  __syncthreads();
  // ---- direction of the thread's own pixels (iu:395-420): rows ty * GN_K .. + GN_K - 1, column tx
  float2 dir[GN_K];
  {
    float w[GN_K + 4][5];
    const float *q = &lab[0][(ty * GN_K + 2) * GN_LW + tx + 2];
#pragma unroll
    for (int r = 0; r < GN_K + 4; r++)
#pragma unroll
      for (int c = 0; c < 5; c++) w[r][c] = q[r * GN_LW + c];
#pragma unroll
    for (int k = 0; k < GN_K; k++) {
      float vx = 0, vy = 0;
#pragma unroll
      for (int yy = -2; yy <= 2; yy++)
#pragma unroll
        for (int xx = -2; xx <= 2; xx++) {
          // (the kernel's middle column / row is zero: adding 0 * s to a sum that started from +0 changes nothing for finite s - the blurred L is)
          const float s = w[k + yy + 2][xx + 2];
          if (xx != 0) vx += v5c((xx + 2) + (yy + 2) * 5) * s;
          if (yy != 0) vy += v5c((yy + 2) + (xx + 2) * 5) * s;
        }
      float len = vx * vx + vy * vy;
      if ((double)len > 1e-10) { len = 1.0f / sqrtf(len); vx *= len; vy *= len; }
      else vx = vy = 0.70710678118f;
      dir[k] = make_float2(vx, vy);
      if (TAPS) { const int x = x0 + tx, y = y0 + ty * GN_K + k; if (x < iw && y < ih) taps.vxy[y * iw + x] = dir[k]; }
    }
  }
  // ---- strength (iu:422-437) of the cells the suppression reads
  const bool interior = x0 >= 3 && x0 + 67 < iw && y0 >= 3 && y0 + GN_ROWS + 3 < ih;
  auto strength_at = [&](int lx, int ly) {      // Lab-tile coordinates of the cell's centre
    float n[3], s[3], w[3], e[3], nw[3], ne[3], sw[3], se[3];
    const int c = ly * GN_LW + lx;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float *p = &lab[1 + k][c];
      nw[k] = p[-GN_LW - 1]; n[k] = p[-GN_LW]; ne[k] = p[-GN_LW + 1];
      w[k] = p[-1]; e[k] = p[1];
      sw[k] = p[GN_LW - 1]; s[k] = p[GN_LW]; se[k] = p[GN_LW + 1];
    }
    return ep_strength(n, s, w, e, nw, ne, sw, se);
  };
  if (interior) {
    // columns 0..63 of the strength tile: a wave takes a band of rows, each lane a column of it (the three rows of a cell serve its neighbours too)
    constexpr int BAND = (GN_SH + GN_TY - 1) / GN_TY;
    const int r0 = ty * BAND, r1 = r0 + BAND < GN_SH ? r0 + BAND : GN_SH;
    {
      float u[BAND + 2][3][3];     // [row][column][channel]
#pragma unroll
      for (int r = 0; r < BAND + 2; r++) {
        const int ly = r0 + r < GN_LH - 1 ? r0 + r : GN_LH - 1;
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
          for (int k = 0; k < 3; k++) u[r][c][k] = lab[1 + k][ly * GN_LW + tx + c];
      }
#pragma unroll
      for (int r = 0; r < BAND; r++)
        if (r0 + r < r1) {
          const float v = ep_strength(u[r][1], u[r + 2][1], u[r + 1][0], u[r + 1][2], u[r][0], u[r][2], u[r + 2][0], u[r + 2][2]);
          str[(r0 + r) * TT_PITCH + tx] = v;
          if (TAPS) { const int sy = r0 + r; if (sy >= 3 && sy < 3 + GN_ROWS && tx >= 3) taps.strength[(y0 - 3 + sy) * iw + x0 - 3 + tx] = v; }
        }
    }
    // columns 64..70: a cell per thread
    for (int t = tid; t < 7 * GN_SH; t += GN_NT) {
      const int sy = t / 7, sx = 64 + t - sy * 7;
      const float v = strength_at(sx + 1, sy + 1);
      str[sy * TT_PITCH + sx] = v;
      if (TAPS) { if (sy >= 3 && sy < 3 + GN_ROWS && sx < 67) taps.strength[(y0 - 3 + sy) * iw + x0 - 3 + sx] = v; }
    }
  } else {
    for (int t = tid; t < GN_SW * GN_SH; t += GN_NT) {
      const int sy = t / GN_SW, sx = t - sy * GN_SW;
      const int fx = x0 - 3 + sx, fy = y0 - 3 + sy;
      if (fx > iw + 3 || fy > ih + 3) continue;          // no pixel of the frame reads this cell
      const int mx = mirror1(fx, iw), my = mirror1(fy, ih);
      const float v = strength_at(mx - (x0 - 4), my - (y0 - 4));
      str[sy * TT_PITCH + sx] = v;
      if (TAPS) { if (sy >= 3 && sy < 3 + GN_ROWS && sx >= 3 && sx < 67 && fx < iw && fy < ih) taps.strength[fy * iw + fx] = v; }
    }
  }
  __syncthreads();
  // ---- suppression (iu:456-471), as k_thinthres: inner samples for every pixel, outer samples for the local maxima only (worked off as a list)
  float4 *lst = (float4 *)&lab[0][0];
  int *lpix = (int *)(lst + 64 * GN_ROWS);
  const int x = x0 + tx;
#pragma unroll
  for (int k = 0; k < GN_K; k++) {
    const int r = ty * GN_K + k;
    const int y = y0 + r;
    const bool inside = x < iw && y < ih;
    bool peak = false;
    float am1 = 0, ap1 = 0;
    const float2 v = dir[k];
    if (inside) {
      const float a0 = str[(r + 3) * TT_PITCH + tx + 3];
      am1 = bicubic_lds(str, x - 1 * v.x, y - 1 * v.y, x0, y0);
      ap1 = bicubic_lds(str, x + 1 * v.x, y + 1 * v.y, x0, y0);
      peak = am1 <= a0 && a0 >= ap1;
      if (!peak) out[y * iw + x] = 0.0f;
    }
    const unsigned long long m = __ballot(peak);
    if (m) {
      const int leader = __ffsll((long long)m) - 1;
      int b = 0;
      if (tx == leader) b = atomicAdd(&nlst, __popcll(m));
      b = __shfl(b, leader);
      if (peak) { const int i = b + __popcll(m & ((1ull << tx) - 1)); lst[i] = make_float4(am1, ap1, v.x, v.y); lpix[i] = r * 64 + tx; }
    }
  }
  __syncthreads();
  const int n = nlst;
  for (int i = tid; i < n; i += GN_NT) {
    const float4 e = lst[i];
    const int c = lpix[i], r = c >> 6, cx = c & 63;
    const int xx = x0 + cx, yy = y0 + r;
    const float a0 = str[(r + 3) * TT_PITCH + cx + 3];
    const float am2 = bicubic_lds(str, xx - 2 * e.z, yy - 2 * e.w, x0, y0);
    const float ap2 = bicubic_lds(str, xx + 2 * e.z, yy + 2 * e.w, x0, y0);
    out[yy * iw + xx] = am2 + e.x + a0 + e.y + ap2;
  }
}

}  // namespace

namespace rdk {

// whether grad_nms() covers a frame of this size (tiles at the right / lower border must hold two pixels for their mirrored cells; tiny frames
// reflect twice): other sizes take edgevec + edge_plab + thinthres
int grad_nms_fits(int iw, int ih) {
  return iw >= 16 && ih >= 16 && (iw % 64 == 0 || iw % 64 >= 2) && (ih % GN_ROWS == 0 || ih % GN_ROWS >= 2);
}
void grad_nms(hipStream_t s, float *nms, float *const bl[3], int iw, int ih, int nz, size_t zs, uint32_t *tap_plab1, float *tap_vxy, float *tap_strength) {
  P3c b = { { bl[0], bl[1], bl[2] } };
  GnTaps t = { tap_plab1, (float2 *)tap_vxy, tap_strength };
  const dim3 grid(rd_tile_blocks(cdiv(iw, 64), cdiv(ih, GN_ROWS), nz));
  const int gdim = rd_gdim(cdiv(iw, 64), cdiv(ih, GN_ROWS), nz);
  if (tap_plab1) hipLaunchKernelGGL(k_grad_nms<1>, grid, dim3(64, GN_TY), 0, s, nms, b, iw, ih, zs, t, gdim);
  else hipLaunchKernelGGL(k_grad_nms<0>, grid, dim3(64, GN_TY), 0, s, nms, b, iw, ih, zs, t, gdim);
}
}  // namespace rdk
