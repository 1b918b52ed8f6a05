// rectdetect-mi355x: rect-path kernels for gfx950 - edge tidy, edge-stopped blur, quantise/despeckle, merge mask,
// region labelling, boundary marking, segment/boundary voting and result sampling.
//
// Reference behaviour being reproduced: oclrect.cl ("rc"), oclrect.c ("rh").  See rd_device.h for arithmetic rules.
#include <stdlib.h>
#include "rd_device.h"
#include "rd_kernels.h"
#include "rd_tidy_tile.h"
#include "rd_poly_scratch.h"
#include "rd_post_core.h"
#include <atomic>

namespace {

using namespace rd;

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
const dim3 block2(64, 4);
inline dim3 grid2(int iw, int ih) { return dim3(cdiv(iw, 64), cdiv(ih, 4)); }
inline int ew_grid(int n) { int g = cdiv(n, 256 * 4); return g < 1 ? 1 : (g > 4096 ? 4096 : g); }

#define RD_XY const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + rd_ty()

// ------------------------------------------------------------------------------------------------ edge tidy
typedef unsigned long long u64;
// A row of a bit plane (ceil(iw / 64) words per row, bit b of word k = pixel 64 k + b) around word k with one cell of margin: bit b = column 64 k - 1 + b
__device__ __forceinline__ bitrow row_m1(const u64 *__restrict__ plane, int wpr, int y, int k, int ih) {
  if (y < 0 || y >= ih) return br(0, 0);
  const u64 *r = plane + (size_t)y * wpr + k;
  const u64 W = k > 0 ? r[-1] : 0ull, C = r[0], E = k + 1 < wpr ? r[1] : 0ull;
  return br((W >> 63) | (C << 1), (C >> 63) | (E << 1));
}
// rc:74-95 on the strong mask's bit plane: a pixel's count is the number of on-pixels of its 3x3 block (1 -> 0; frame border 0).  All the merge mask
// asks of the counts is "counted" (count != 0: on, with an on-neighbour) and "curve end" (count == 2: exactly one on-neighbour): two bit rows per
// 64 pixels, words [0] and [1] of bits[(y * wpr + k) * 2].  One thread per word.  (The counts themselves - what the region sizes start from, quirk
// H2 - are evaluated from the same bit plane where they are used: k_region_init.)
__global__ __launch_bounds__(256) void k_junction_bits(u64 *__restrict__ bits, const u64 *__restrict__ strong, int iw, int ih, int wpr, size_t zs) {
  RD_ZSHIFT(zs, bits, strong);
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= wpr * ih) return;
  const int y = t / wpr, k = t - y * wpr;
  const bitrow u = row_m1(strong, wpr, y - 1, k, ih), m = row_m1(strong, wpr, y, k, ih), d = row_m1(strong, wpr, y + 1, k, ih);
  bitrow ge1, ge2;
  br_count8(u, m, d, ge1, ge2);
  const bitrow in1 = (y >= 1 && y <= ih - 2) ? br_range(1 - 64 * k + 1, iw - 2 - 64 * k + 1) : br(0, 0);
  const bitrow any = m & in1 & ge1, end = any & ~ge2;
  bits[(size_t)t * 2] = (any.lo >> 1) | (any.hi << 63);
  bits[(size_t)t * 2 + 1] = (end.lo >> 1) | (end.hi << 63);
}

// ------------------------------------------------------------------------------------------------ edge-stopped box blur
// rc:155-205: mean of the packed-Lab integer fields over up to 4 pixels on either side along one axis (centre counted
// twice), each side stopping at transitions of the int8 edge mask.  The stopping positions depend on the mask only and
// the mask is the same for all 20 passes (rh:286-296), so they are computed once per frame (k_blblur_extents: number of
// samples taken towards smaller / larger coordinates, 0..5 each, for both axes) and every pass is a branch-free gather.
//
// The scans only ask "is the mask non-zero" of cells up to 5 away along the axis and 1 sideways, so the block turns its
// (64+16) x (BE_ROWS+11) patch of the mask into bit rows (one ballot per 64 cells) and every stopping rule into one word
// operation per row:
//   towards smaller coordinates, cell c stops the scan before it is counted when
//       c < 0, or c > 0 and (E[c] & !E[c-1]  |  !E[c] & E[c-1] & E[c, one step sideways]);
//   towards larger coordinates when
//       c > n-1, or (centre on an edge ? !E[c] : !E[c] & E[c+1]).
// Cells outside the frame are staged as zero, which makes the reference's `c < n-1` and `has side cell` tests redundant.
// A pixel then reads its five relevant stop bits per direction and counts the leading clear ones.
#ifndef BE_ROWS
#define BE_ROWS 32
#endif
#define BE_NR (BE_ROWS + 11)          // rows y0-5 .. y0+BE_ROWS+5
__global__ __launch_bounds__(256) void k_blblur_extents(uint16_t *__restrict__ ext, const int8_t *__restrict__ edge, int iw, int ih, size_t zs) {
  RD_ZSHIFT(zs, ext, edge);
  typedef unsigned long long u64;
  __shared__ u64 EA[BE_NR + 1], EB[BE_NR + 1];   // mask != 0 for columns x0-8 .. x0+55 (A) and x0+56 .. x0+71 (B)
  __shared__ u64 HLa[BE_NR], HLb[BE_NR], HRa[BE_NR], HRb[BE_NR];   // along x: stop bits towards smaller / larger x (centre not on an edge)
  __shared__ u64 VL[BE_NR], VR[BE_NR], VE[BE_NR];                  // along y, tile columns only (bit = column - x0)
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * BE_ROWS;
  const int tx = threadIdx.x, ty = rd_ty(), tid = ty * 64 + tx;
  {
    constexpr int IT = (BE_NR + 3) / 4;
    int8_t va[IT], vb[IT];
    bool oka[IT], okb[IT];
    const int xa = x0 - 8 + tx, xb = x0 + 56 + tx;
#pragma unroll
    for (int i = 0; i < IT; i++) {          // all loads first (see stage_cells)
      const int yy = y0 - 5 + ty + 4 * i;
      const bool rowin = ty + 4 * i < BE_NR && yy >= 0 && yy < ih;
      oka[i] = rowin && xa >= 0 && xa < iw;
      okb[i] = rowin && tx < 16 && xb < iw;
      va[i] = edge[oka[i] ? yy * iw + xa : 0];
      vb[i] = edge[okb[i] ? yy * iw + xb : 0];
    }
#pragma unroll
    for (int i = 0; i < IT; i++) {
      const int r = ty + 4 * i;
      const u64 ba = __ballot(oka[i] && va[i] != 0), bb = __ballot(okb[i] && vb[i] != 0);
      if (tx == 0 && r < BE_NR) { EA[r] = ba; EB[r] = bb; }
    }
  }
  if (tid == 0) { EA[BE_NR] = 0; EB[BE_NR] = 0; }
  __syncthreads();
  if (tid < BE_NR) {
    const int r = tid, yy = y0 - 5 + r;
    const u64 A = EA[r], B = EB[r], SA = EA[r + 1], SB = EB[r + 1];
    const u64 PA = A << 1, PB = (B << 1) | (A >> 63);            // E[c-1]
    const u64 NA = (A >> 1) | (B << 63), NB = B >> 1;            // E[c+1]
    u64 la = (A & ~PA) | (~A & PA & SA), lb = (B & ~PB) | (~B & PB & SB);
    u64 ra = ~A & NA, rb = ~B & NB;
    if (x0 == 0) { la &= ~(1ull << 8); la |= 0xffull; }          // column 0 never stops the scan itself; columns < 0 do
    const int fa = iw - x0 + 8, fb = iw - x0 - 56;               // first bit index beyond the frame
    if (fa < 64) ra |= ~0ull << fa;
    if (fb <= 0) rb = ~0ull; else if (fb < 64) rb |= ~0ull << fb;
    HLa[r] = la; HLb[r] = lb; HRa[r] = ra; HRb[r] = rb;
    const u64 Et = (A >> 8) | (B << 56), En = (A >> 9) | (B << 55);                 // columns x0.., x0+1..
    const u64 Ep = r > 0 ? (EA[r - 1] >> 8) | (EB[r - 1] << 56) : 0ull;             // row above
    const u64 Es = (SA >> 8) | (SB << 56);                                           // row below
    u64 vl = (Et & ~Ep) | (~Et & Ep & En);
    if (yy == 0) vl = 0;
    if (yy < 0) vl = ~0ull;
    u64 vr = ~Et & Es;
    if (yy >= ih) vr = ~0ull;
    VL[r] = vl; VR[r] = vr; VE[r] = Et;
  }
  __syncthreads();
  const int x = x0 + tx;
  if (x >= iw) return;
  for (int rr = ty; rr < BE_ROWS; rr += 4) {
    const int y = y0 + rr;
    if (y >= ih) break;
    const int r = rr + 5;
    // 64-bit windows starting at column x - 8: the pixel is bit 8
    const int sh = tx, rs = (64 - tx) & 63;
    const u64 keep = tx == 0 ? 0ull : ~0ull;
    const u64 we = (EA[r] >> sh) | ((EB[r] << rs) & keep);
    const u64 wl = (HLa[r] >> sh) | ((HLb[r] << rs) & keep);
    const u64 wr1 = (HRa[r] >> sh) | ((HRb[r] << rs) & keep);
    const bool oe = (we >> 8) & 1;
    const u64 wr = oe ? ~we : wr1;
    const unsigned tl = (unsigned)(wl >> 4) & 31u;                 // bit 4: the pixel's own column ... bit 0: four columns before
    const int nl = __clz((int)tl) - 27;
    const int nr = __ffs((int)(((unsigned)(wr >> 8) & 31u) | 32u)) - 1;
    unsigned tv = 0, uv = 32u;
#pragma unroll
    for (int d = 0; d < 5; d++) {
      tv |= (unsigned)((VL[r - d] >> tx) & 1ull) << (4 - d);
      const u64 w = oe ? ~VE[r + d] : VR[r + d];
      uv |= (unsigned)((w >> tx) & 1ull) << d;
    }
    const int nlv = __clz((int)tv) - 27, nrv = __ffs((int)uv) - 1;
    // (a pixel without any run keeps its value: written as "the centre alone" - 0 samples towards smaller, 1 towards larger coordinates - which the passes'
    //  sum / count reproduces exactly, so that they need no special case)
    const int nrh = (nl | nr) ? nr : 1, nrvv = (nlv | nrv) ? nrv : 1;
    ext[y * iw + x] = (uint16_t)((unsigned)(nl | (nrh << 3)) | ((unsigned)(nlv | (nrvv << 3)) << 6));
  }
}

// floor(s / w) for a sum s of w samples of a field (s <= 4095 w, 1 <= w <= 10) as one 24-bit multiplication and a shift: s * ceil(2^19 / w) >> 19
// (exhaustively checked: tools/check_div_small.py; the product stays below 2^32.  Before: float(s) * (1/w) + 0.5 * (1/w) through a conversion, an
// fma and a conversion back - three operations per field instead of two)
__device__ __forceinline__ unsigned div_small_m(unsigned s, unsigned m) { return __umul24(s, m) >> 19; }

// One (horizontal, vertical) pair of passes (rh:286-296) in a single launch, with RUNNING SUMS along the axis instead of ten samples per pixel (the form of
// rounds 1-3 and most of round 4 staged the tile in LDS and read a pixel's ten samples - five per side, the ones beyond the run from a region of zeros -
// with a compare and a select per sample: 59 vector instructions per pixel and pass, 70.0 us per launch of 8 frames against 53.4).
// A window's sum is a difference of two prefix values plus the
// centre:  sum = P[c + nr] - P[c + 1 - nl] + v[c]  (P[i] = sum of the cells before i; covers nl = 0, nr = 0 and both), and the prefix words are plain 32-bit
// sums of the expanded cells (L | a << 16, b): whatever a prefix wrapped to or carried between its fields, the DIFFERENCE is the window's own sum, whose
// fields stay below 2^16 - exact.  That takes the ten selected LDS reads, ten compares, ten selects and ten additions of a pixel down to four reads and six
// subtractions; what it costs is building the prefixes, which is arranged so that the thread that produces a cell also holds its neighbours along the axis:
//   horizontal: lanes = the tile's 62 rows, wave s walks columns 9s .. 9s+8 of its row with the sum in registers (row pitch 73: conflict-free), the
//               segment totals meet in LDS, every thread adds the totals of the segments before its own and writes its nine prefix values in place;
//   vertical:   lanes = columns, wave w computes the horizontal results of rows 8w .. 8w+7 (from the row prefixes) and sums them on the way; totals and
//               offsets as before; the column prefixes take the memory the row prefixes had.
// Tile: 64 x BQ_ROWS outputs (BQ_ROWS + 8 <= 64 rows of the horizontal strip are the lanes of a wave), 512 threads, 40 KB of LDS and 60 registers: four blocks per CU.
#ifndef BQ_ROWS
#define BQ_ROWS 54            // (1080 = 20 x 54)
#endif
#define BQ_HR (BQ_ROWS + 8)   // rows of the staged tile / horizontal strip
#define BQ_SW 73
static_assert(BQ_ROWS + 9 <= 64, "the column prefixes have 64 rows: the last output row reads prefix row BQ_ROWS + 8");
#ifndef BQ_WB
#define BQ_WB 2              // pixels of a thread evaluated together (1 / 2 / 4 / 8: 53.4 / 53.1 / 54.0 / 62.5 us per launch of 8 frames - 8 costs a wave per SIMD)
#endif
__global__ __launch_bounds__(512) void k_blblur_pair(uint32_t *__restrict__ out, const uint16_t *__restrict__ ext, const uint32_t *__restrict__ in, int iw, int ih, size_t zs, int gdim) {
  const rd_tile rd_b = rd_block_tile(gdim);
  if (rd_b.x < 0) return;
  RD_ZSHIFTZ(rd_b.z, zs, out, ext, in);
  typedef unsigned long long u64;             // a cell, a prefix, a window sum: (L | a << 16) in the low word, b in the high word - one 64-bit addition per step
  __shared__ u64 pl[BQ_HR * BQ_SW > 64 * 64 ? BQ_HR * BQ_SW : 64 * 64];      // staged cells -> row prefixes (73 per row) -> column prefixes (64 rows x 64); 40 KB with the totals: four blocks per CU
#ifndef BQ_PAD
#define BQ_PAD 0
#endif
  __shared__ u64 tot[8 * 64 + BQ_PAD];        // segment totals of the scan in progress
  __shared__ unsigned rwt[16];
  // (the wave's index as a SCALAR: everything that depends on it alone - row numbers, row addresses, "this row does not exist" - stays out of the vector unit)
  const int tx = threadIdx.x, wv = rd_ty(), tid = wv * 64 + tx;
  const int x0 = rd_b.x * 64, y0 = rd_b.y * BQ_ROWS;
  const int x = x0 + tx;
  constexpr int NV = (BQ_ROWS + 7) / 8;
  unsigned eh[8], ev[NV];
  uint32_t q[9];
  // staging: wave wv takes rows wv + 8 i of the tile, its lanes columns 0..63; the eight columns left over are a cell per thread (row tid / 8) - no division,
  // row addresses advance by a scalar.  Run extents of the thread's pixels: rows 8 wv .. 8 wv + 7 of the horizontal strip, rows wv + 8 k of the output tile.
  // Every load is unconditional (clamped address, value replaced afterwards) and requested before the first is used.
  const int sr = tid >> 3, sc = 64 + (tid & 7);
  const bool interior = x0 >= 4 && y0 >= 4 && x0 + 68 <= iw && y0 + BQ_ROWS + 4 <= ih;
  if (interior) {
    const unsigned ib = (unsigned)((y0 - 4 + wv) * iw + x0 - 4 + tx);
#pragma unroll
    for (int i = 0; i < 8; i++) q[i] = wv + 8 * i < BQ_HR ? atu(in, ib + (unsigned)(8 * i * iw)) : 0u;
    q[8] = sr < BQ_HR ? atu(in, (unsigned)((y0 - 4) * iw + x0 - 4) + (unsigned)__umul24(sr, iw) + (unsigned)sc) : 0u;
    const unsigned eb = (unsigned)((y0 - 4 + 8 * wv) * iw + x);
#pragma unroll
    for (int j = 0; j < 8; j++) eh[j] = 8 * wv + j < BQ_HR ? (unsigned)atu(ext, eb + (unsigned)(j * iw)) : 0u;
    const unsigned vb = (unsigned)((y0 + wv) * iw + x);
#pragma unroll
    for (int k = 0; k < NV; k++) ev[k] = wv + 8 * k < BQ_ROWS ? (unsigned)atu(ext, vb + (unsigned)(8 * k * iw)) : 0u;
  } else {
    bool okq[9], okh[8], okv[NV];
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const int r = i < 8 ? wv + 8 * i : sr, c = i < 8 ? tx : sc;
      const int xx = x0 - 4 + c, yy = y0 - 4 + r;
      okq[i] = r < BQ_HR && xx >= 0 && xx < iw && yy >= 0 && yy < ih;
      q[i] = atu(in, okq[i] ? (unsigned)(yy * iw + xx) : 0u);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int r = 8 * wv + j, y = y0 - 4 + r;
      okh[j] = r < BQ_HR && x < iw && y >= 0 && y < ih;
      eh[j] = atu(ext, okh[j] ? (unsigned)(y * iw + x) : 0u);
    }
#pragma unroll
    for (int k = 0; k < NV; k++) {
      const int r = wv + 8 * k, y = y0 + r;
      okv[k] = r < BQ_ROWS && x < iw && y < ih;
      ev[k] = atu(ext, okv[k] ? (unsigned)(y * iw + x) : 0u);
    }
#pragma unroll
    for (int i = 0; i < 9; i++) if (!okq[i]) q[i] = 0u;
#pragma unroll
    for (int j = 0; j < 8; j++) if (!okh[j]) eh[j] = 0u;
#pragma unroll
    for (int k = 0; k < NV; k++) if (!okv[k]) ev[k] = 0u;
  }
  if (tid < 16) rwt[tid] = tid >= 1 && tid <= 10 ? ((1u << 19) + (unsigned)tid - 1u) / (unsigned)tid : 0u;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int r = i < 8 ? wv + 8 * i : sr, c = i < 8 ? tx : sc;
    const uint32_t v = q[i];
    if (r < BQ_HR) pl[r * BQ_SW + c] = (u64)((v & 4095u) | ((v << 4) & 0x3ff0000u)) | ((u64)(v >> 22) << 32);
  }
  __syncthreads();
  // ---- row prefixes: lane = row, wave = segment of nine columns
  {
    u64 p[10], v[9], sum = 0;
    const int b = (tx < BQ_HR ? tx : BQ_HR - 1) * BQ_SW + 9 * wv;
#pragma unroll
    for (int j = 0; j < 9; j++) v[j] = pl[b + j];
#pragma unroll
    for (int j = 0; j < 9; j++) { p[j] = sum; sum += v[j]; }
    p[9] = sum;
    tot[wv * 64 + tx] = sum;
    __syncthreads();
    u64 off = 0, ts[7];
#pragma unroll
    for (int s2 = 0; s2 < 7; s2++) ts[s2] = s2 < wv ? tot[s2 * 64 + tx] : 0;      // (wv is scalar: only the totals that count are read, all of them in flight together)
#pragma unroll
    for (int s2 = 0; s2 < 7; s2++) off += ts[s2];
    if (tx < BQ_HR) {      // (lanes 62, 63 have no row)
#pragma unroll
      for (int j = 0; j < 9; j++) pl[b + j] = p[j] + off;
      if (wv == 7) pl[b + 9] = p[9] + off;      // P[72]
    }
  }
  __syncthreads();
  // ---- horizontal results of rows 8 wv .. 8 wv + 7 (lane = column), summed down the column on the way
  // (BQ_WB pixels at a time: all their reads are requested before the first is used.  No pixel of the frame has an empty window - k_blblur_extents turns
  //  "no run at all" into "the centre alone", which divides to itself; cells outside the frame carry extents 0 and weight 0 and come out as the zeros they are)
  u64 pv[9];
  {
    u64 sum = 0;
#pragma unroll
    for (int j0 = 0; j0 < 8; j0 += BQ_WB) {
      u64 hi[BQ_WB], lo[BQ_WB], c0[BQ_WB], c1[BQ_WB];
      unsigned rw[BQ_WB];
#pragma unroll
      for (int u = 0; u < BQ_WB; u++) {
        const unsigned e = eh[j0 + u];
        const int nl = e & 7, nr = (e >> 3) & 7;
        const int c = (8 * wv + j0 + u < BQ_HR ? 8 * wv + j0 + u : BQ_HR - 1) * BQ_SW + tx + 4;      // (rows 62, 63 of wave 7 do not exist: any address, the result is dropped)
        hi[u] = pl[c + nr]; lo[u] = pl[c + 1 - nl]; c0[u] = pl[c]; c1[u] = pl[c + 1];
        rw[u] = rwt[nl + nr];
      }
#pragma unroll
      for (int u = 0; u < BQ_WB; u++) {
        const u64 a = (hi[u] + c1[u]) - (lo[u] + c0[u]);
        const unsigned ax = (unsigned)a, ay = (unsigned)(a >> 32);
        u64 o = (u64)(div_small_m(ax & 0xffffu, rw[u]) | (div_small_m(ax >> 16, rw[u]) << 16)) | ((u64)div_small_m(ay, rw[u]) << 32);
        if (8 * wv + j0 + u >= BQ_HR) o = 0;
        pv[j0 + u] = sum;
        sum += o;
      }
    }
    pv[8] = sum;
  }
  tot[wv * 64 + tx] = pv[8];       // (the totals of the row scan were last read before the barrier above)
  __syncthreads();                 // (every read of the row prefixes is done as well: their memory takes the column prefixes)
  {
    u64 off = 0, ts[7];
#pragma unroll
    for (int s2 = 0; s2 < 7; s2++) ts[s2] = s2 < wv ? tot[s2 * 64 + tx] : 0;
#pragma unroll
    for (int s2 = 0; s2 < 7; s2++) off += ts[s2];
#pragma unroll
    for (int j = 0; j < 8; j++) pl[(8 * wv + j) * 64 + tx] = pv[j] + off;
  }
  __syncthreads();
  if (x >= iw) return;
#pragma unroll
  for (int k0 = 0; k0 < NV; k0 += BQ_WB) {
    u64 hi[BQ_WB], lo[BQ_WB], c0[BQ_WB], c1[BQ_WB];
    unsigned rw[BQ_WB];
#pragma unroll
    for (int u = 0; u < BQ_WB; u++) {
      if (k0 + u >= NV) continue;
      const unsigned e = ev[k0 + u] >> 6;
      const int nl = e & 7, nr = (e >> 3) & 7;
      const int r = wv + 8 * (k0 + u);
      const int c = ((r < BQ_ROWS ? r : 0) + 4) * 64 + tx;
      hi[u] = pl[c + nr * 64]; lo[u] = pl[c + (1 - nl) * 64]; c0[u] = pl[c]; c1[u] = pl[c + 64];
      rw[u] = rwt[nl + nr];
    }
#pragma unroll
    for (int u = 0; u < BQ_WB; u++) {
      if (k0 + u >= NV) continue;
      const int r = wv + 8 * (k0 + u);
      const int y = y0 + r;
      const u64 a = (hi[u] + c1[u]) - (lo[u] + c0[u]);
      const unsigned ax = (unsigned)a, ay = (unsigned)(a >> 32);
      if (r < BQ_ROWS && y < ih) atu(out, (unsigned)(y * iw + x)) = div_small_m(ax & 0xffffu, rw[u]) | (div_small_m(ax >> 16, rw[u]) << 12) | (div_small_m(ay, rw[u]) << 22);
    }
  }
}

// rc:207-216: each Lab field rounded to n levels
__device__ __forceinline__ uint32_t quantize_plab(uint32_t v, int n0, int n1, int n2) {
  float L, a, b;
  unpack_lab(v, L, a, b);
  return pack_lab(roundf(L * n0) / (float)n0, roundf(a * n1) / (float)n1, roundf(b * n2) / (float)n2);
}

// Quantising a packed Lab word field by field costs a rounding and an IEEE division per field; each result depends on nothing but
// the field's 12 / 10 bits, so the 24-level tables are built once per device by the very function above (k_quant24_lut) and the
// frame path looks them up: g_quant24[0..4095] = quantised L field, [4096..5119] = quantised a / b field.
__device__ __attribute__((aligned(16))) uint16_t g_quant24[4096 + 1024];     // (read through 32-bit loads in k_despeckle<24>)
__global__ void k_quant24_lut() {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 4096) g_quant24[i] = (uint16_t)(quantize_plab((uint32_t)i, 24, 24, 24) & 4095u);
  else if (i < 5120) g_quant24[i] = (uint16_t)((quantize_plab((uint32_t)(i - 4096) << 12, 24, 24, 24) >> 12) & 1023u);      // (a and b share the formula)
}

// rc:218-244: pixels with a non-zero NMS response take the colour of the Lab-nearest 3x3 neighbour without one.
// One block per 64 x DS_ROWS tile: the tile and a 1-cell halo of both inputs are staged in LDS (colours already quantised,
// the response reduced to flags) with all loads of a thread in flight together; the affected pixels - a few per cent, on thin
// lines that cross a third of all waves - are collected in an LDS list and worked off with all lanes busy, from LDS only.
#ifndef DS_ROWS
#define DS_ROWS 16
#endif
#define DS_P 66
// QN > 0: the input is quantised to QN levels per field on the fly (rc:207-216 fused in: no separate pass over the plane)
template <int QN>
__global__ __launch_bounds__(256) void k_despeckle(uint32_t *__restrict__ out, const uint32_t *__restrict__ in, const float *__restrict__ edge, int iw, int ih, size_t zs, int gdim) {
  const rd_tile rd_b = rd_block_tile(gdim);
  if (rd_b.x < 0) return;
  RD_ZSHIFTZ(rd_b.z, zs, out, in, edge);
  constexpr int NC = (DS_ROWS + 2) * DS_P, IT = (NC + 255) / 256;
  __shared__ uint32_t tq[IT * 256];      // (padded to whole rounds of the block: the staging below is straight-line code - no `cell exists` branch for the compiler to sink a load into)
  __shared__ uint8_t tf[IT * 256];        // bit 0: replaced as a centre (!(e < 1e-6)), bit 1: skipped as a neighbour (e >= 1e-6), bit 2: outside the frame
  __shared__ int list[64 * DS_ROWS];
  __shared__ int nlist;
  __shared__ uint32_t qlut[QN == 24 ? 2560 : 1];      // g_quant24, two entries per word
  const int tx = threadIdx.x, tid = rd_ty() * 64 + tx;
  const int x0 = rd_b.x * 64, y0 = rd_b.y * DS_ROWS;
  if (tid == 0) nlist = 0;
  {
    // the tile's cells and the quantisation tables are requested together (the cells were loaded a pair at a time before, each pair behind the previous
    // one's stores - the compiler had sunk every load into the branch that used it: five trips to memory one after the other, after the tables' one)
    uint32_t v[IT];
    float e[IT];
    bool ok[IT];
#pragma unroll
    for (int i = 0; i < IT; i++) {
      const int t = tid + 256 * i;
      const int xx = x0 - 1 + t % DS_P, yy = y0 - 1 + t / DS_P;
      ok[i] = t < NC && xx >= 0 && xx < iw && yy >= 0 && yy < ih;
      const int a = ok[i] ? yy * iw + xx : 0;
      v[i] = in[a];
      e[i] = edge[a];
    }
    if (QN == 24) {
      uint32_t w[10];
#pragma unroll
      for (int i = 0; i < 10; i++) w[i] = ((const uint32_t *)g_quant24)[tid + 256 * i];
#pragma unroll
      for (int i = 0; i < 10; i++) qlut[tid + 256 * i] = w[i];
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < IT; i++) {
      const int t = tid + 256 * i;
      uint32_t qv = v[i];
      if (QN == 24) {
        const uint16_t *q16 = (const uint16_t *)qlut;
        qv = (uint32_t)q16[qv & 4095u] | ((uint32_t)q16[4096 + ((qv >> 12) & 1023u)] << 12) | ((uint32_t)q16[4096 + (qv >> 22)] << 22);
      } else if (QN > 0) qv = quantize_plab(qv, QN, QN, QN);
      tq[t] = ok[i] ? qv : 0u;
      tf[t] = ok[i] ? (uint8_t)((!(e[i] < 1e-6f) ? 1 : 0) | (e[i] >= 1e-6f ? 2 : 0)) : (uint8_t)4;
    }
  }
  __syncthreads();
  for (int r = rd_ty(); r < DS_ROWS; r += 4) {
    const int x = x0 + tx, y = y0 + r;
    const int i = (r + 1) * DS_P + tx + 1;
    bool hot = false;
    if (x < iw && y < ih) {
      hot = tf[i] & 1;
      if (!hot) out[y * iw + x] = tq[i];
    }
    const unsigned long long m = __ballot(hot);
    if (m) {
      const int leader = __ffsll((long long)m) - 1;
      int o = 0;
      if (tx == leader) o = atomicAdd(&nlist, __popcll(m));
      o = __shfl(o, leader);
      if (hot) list[o + __popcll(m & ((1ull << tx) - 1))] = i;
    }
  }
  __syncthreads();
  const int n = nlist;
  for (int j = tid; j < n; j += 256) {
    const int i = list[j];
    uint32_t r = tq[i];
    float dist = 1e+10f, l0, a0, b0;
    unpack_lab(r, l0, a0, b0);
#pragma unroll
    for (int k = 0; k < 9; k++) {
      const int q = i + (k / 3 - 1) * DS_P + k % 3 - 1;
      if (tf[q] & 6) continue;          // outside the frame, or has a response itself
      float l1, a1, b1;
      const uint32_t v = tq[q];
      unpack_lab(v, l1, a1, b1);
      const float dx = l1 - l0, dy = a1 - a0, dz = b1 - b0;
      const float d = sqrtf(dx * dx + dy * dy + dz * dz);
      if (d < dist) { r = v; dist = d; }
    }
    out[(y0 + i / DS_P - 1) * iw + x0 + i % DS_P - 1] = r;
  }
}

// ------------------------------------------------------------------------------------------------ merge mask
// rc:246-287.  The reference scatters: every pixel with a non-zero junction count sets the ring 16 <= d^2 < 36 around
// itself, then curve ends (count 2) erase the disc d^2 < 64 and all other counted pixels the disc d^2 < 16.  Equivalent
// gather: mask(p) = A & ~B & ~C with A = "a counted pixel lies on the ring around p", B = "a curve end within d^2 < 64",
// C = "another counted pixel within d^2 < 16".  The pixel classes arrive as bit rows (k_junction_bits) and the mask leaves as a bit plane
// (ceil(iw / 64) words per row): its only reader, k_region_init, wants one bit per pixel.
// The 64 cells of a word seeing the cell DX columns to their right (DX < 0: to their left): w0 w1 w2 = the words left of, at and
// right of the output word in the source row.
template <int DX>
__device__ __forceinline__ unsigned long long mm_shift(unsigned long long w0, unsigned long long w1, unsigned long long w2) {
  if (DX == 0) return w1;
  if (DX > 0) return (w1 >> DX) | (w2 << (64 - DX));
  return (w1 << -DX) | (w0 >> (64 + DX));
}
// OR over dx in [-8, 8] with dx^2 + DY^2 in [LO, HI) of the row shifted by dx (all of it resolved at compile time)
template <int DY, int LO, int HI, int DX = -8>
__device__ __forceinline__ unsigned long long mm_row(unsigned long long w0, unsigned long long w1, unsigned long long w2) {
  unsigned long long r = 0;
  if (DX * DX + DY * DY >= LO && DX * DX + DY * DY < HI) r = mm_shift<DX>(w0, w1, w2);
  if constexpr (DX < 8) r |= mm_row<DY, LO, HI, DX + 1>(w0, w1, w2);
  return r;
}
template <int DY>
__device__ __forceinline__ void mm_rows(const unsigned long long *sb, int r, unsigned long long &A, unsigned long long &B, unsigned long long &C) {
  const unsigned long long *row = sb + (r + 8 + DY) * 6;       // rows outside the frame hold zeros
  const unsigned long long a0 = row[0], a1 = row[1], a2 = row[2], e0 = row[3], e1 = row[4], e2 = row[5];
  constexpr int ADY = DY < 0 ? -DY : DY;
  if (ADY * ADY < 36) A |= mm_row<ADY, 16, 36>(a0, a1, a2);
  if (ADY * ADY < 64) B |= mm_row<ADY, 0, 64>(e0, e1, e2);
  if (ADY * ADY < 16) C |= mm_row<ADY, 0, 16>(a0 & ~e0, a1 & ~e1, a2 & ~e2);
  if constexpr (DY < 8) mm_rows<DY + 1>(sb, r, A, B, C);
}

// block: 64 columns x MM_ROWS rows.  The (MM_ROWS + 16) x 3 words x 2 classes of bit rows it needs are staged in LDS; ONE thread per
// output row then evaluates the three tests for all 64 columns of its row at once - every (dy, dx) of a test is one shifted OR of
// a 64-bit word instead of a 17-bit window test per pixel and row offset (262 -> 45 vector instructions per pixel).
#define MM_ROWS 64
__global__ __launch_bounds__(256) void k_mm_gather(unsigned long long *__restrict__ out, const unsigned long long *__restrict__ bits, int iw, int ih, int wpr, size_t zs) {
  RD_ZSHIFT(zs, out, bits);
  __shared__ unsigned long long sb[(MM_ROWS + 16) * 6];
  const int k = blockIdx.x;                 // word holding this block's own 64 columns
  const int y0 = blockIdx.y * MM_ROWS;
  const int tid = rd_ty() * 64 + threadIdx.x;
  for (int t = tid; t < (MM_ROWS + 16) * 6; t += 256) {
    const int r = t / 6, j = t % 6;         // j: word k-1, k, k+1 of class any (0..2) / end (3..5)
    const int yy = y0 - 8 + r, kk = k - 1 + j % 3;
    unsigned long long v = 0ull;
    if (yy >= 0 && yy < ih && kk >= 0 && kk < wpr) v = bits[((size_t)yy * wpr + kk) * 2 + j / 3];
    sb[t] = v;
  }
  __syncthreads();
  if (tid < MM_ROWS && y0 + tid < ih) {
    unsigned long long A = 0, B = 0, C = 0;
    mm_rows<-8>(sb, tid, A, B, C);
    const int left = iw - 64 * k;           // columns of this word inside the frame
    out[(size_t)(y0 + tid) * wpr + k] = A & ~B & ~C & (left >= 64 ? ~0ull : ((1ull << left) - 1ull));
  }
}

// ------------------------------------------------------------------------------------------------ regions
// rc:289-298 initial links (up if same colour, else left if same colour, else self) - exactly the reference's, chains and all - AND the
// first launch of the merge kernel on them (see k_region_round for the rule), in one tile kernel without a single label load or atomic:
// a raw link reaches one pixel up or left, so everything the first launch does to a pixel t is decided within 10 pixels of it -
//   label1[t] = min( link[t],  G(t),  G(t + 1) if link[t + 1] == t,  G(t + iw) if link[t + iw] == t ),
//   G(p) = the label pixel p proposes (smallest label among itself and the neighbours it may adopt from, then 8 jumps along the links),
//          if p is processed (not on the frame's ring) and that differs from link[p]
// (a proposal goes to the pixel itself and to its link's target, and only t, its right and its lower neighbour can link to t).
// The tile loads colours for its cells plus 10 rows above / columns left (the jumps' reach) and 2 below / right (the neighbours' proposals).
// Also written: one byte per pixel telling from which of its 4 neighbours the pixel may adopt a label in every launch (rc:308-326 - colours,
// merge mask and edges do not change): bit0 up, bit1 left, bit2 right, bit3 down, bit4 processed; 0 for frame-border pixels.
// The labels leave as the words of k_region_round (label << 3, no mark) in both planes its launches alternate between.
#define RR_NFLAGS (RD_REGION_STATUS_AT + 32)      // ints in front of the allow bytes: [0, RD_REGION_MAX_LAUNCHES) one flag per launch of the merge, [RD_REGION_STATUS_AT, + 8) status words of the absorption
static_assert(RD_REGION_STATUS_AT >= RD_REGION_MAX_LAUNCHES && RR_NFLAGS <= 256, "k_region_init clears the flag words with one block of 256 threads");
#ifndef RI_ROWS
#define RI_ROWS 32
#endif
#define RI_H 10                         // halo above / left
#define RI_RW (64 + RI_H + 2)
#define RI_RH (RI_ROWS + RI_H + 2)
#define RI_NC (RI_RW * RI_RH)
// mask / edge: the merge mask and the strong mask as bit planes (wpr words per row); size_out (optional) <- the junction counts of the strong mask
// (rc:74-95), which the region sizes start from (quirk H2: the reference counts into the plane that still holds them)
__global__ __launch_bounds__(256) void k_region_init(int *__restrict__ A, int *__restrict__ B, u64 *__restrict__ allow, const int *__restrict__ pix, const u64 *__restrict__ mask,
                                                     const u64 *__restrict__ edge, int iw, int ih, int wpr, int *__restrict__ flags, int *__restrict__ size_out, size_t zs, int gdim) {
  const rd_tile rd_b = rd_block_tile(gdim);
  if (rd_b.x < 0) return;
  RD_ZSHIFTZ(rd_b.z, zs, A, B, allow, pix, mask, edge, flags, size_out);
  __shared__ u64 sm[(RI_ROWS + 2) * 2];        // merge mask: rows y0 .. y0 + RI_ROWS + 1, words k and k + 1
  __shared__ u64 se[(RI_ROWS + 3) * 3];        // strong mask: rows y0 - 1 .. y0 + RI_ROWS + 1, words k - 1, k, k + 1
  __shared__ int col[RI_NC];                  // colours, then (in place) nothing: kept for the allow bits
  __shared__ short lnk[RI_NC];                // raw link as a cell index of this tile's region (-1: cell outside the frame)
  __shared__ short prop[RI_NC];               // G as a cell index, or 0x7fff
  __shared__ unsigned char alw[RI_NC];
  // (also: the round flags start at zero - flags[0] = 1: the first launch, evaluated here, counts as one that changed something - and the
  //  size plane starts from size_init - quirk H2 - without extra launches)
  const int tid = rd_ty() * 64 + threadIdx.x;
  if (rd_b.x == 0 && rd_b.y == 0 && tid < RR_NFLAGS) flags[tid] = tid == 0 ? 1 : 0;
  const int gx0 = rd_b.x * 64 - RI_H, gy0 = rd_b.y * RI_ROWS - RI_H;
  // colours of the region (cells outside the frame: marked by lnk = -1 below), and - for the tile and two more rows / columns - "merge mask
  // set" and "strong edge" as two bits (all loads of a thread in flight together)
  {
    // wave w takes rows w + 4 i of the region, its lanes columns 0..63; the 12 columns left over are cells (row t >> 4, column 64 + (t & 15)) of a 16-wide
    // strip: no division, a scalar row address, all loads of a thread in flight together
    constexpr int NR = (RI_RH + 3) / 4, NS = (RI_RH * 16 + 255) / 256;
    const int wv = rd_ty(), lx = threadIdx.x;
    int v[NR + NS];
    bool ok[NR + NS];
    // (the words of the two bit planes travel with the colours: unconditional loads of clamped addresses, stored further down - requested after the colours'
    //  stores, each behind a branch, they were two more trips to memory one after the other)
    u64 wm = 0, we = 0;
    bool okm = false, oke = false;
    {
      const int k = rd_b.x, yb0 = rd_b.y * RI_ROWS;
      { const int r = tid >> 1, kk = k + (tid & 1), yy = yb0 + r; okm = tid < (RI_ROWS + 2) * 2 && yy < ih && kk < wpr; wm = mask[okm ? (size_t)yy * wpr + kk : 0]; }
      { const int r = tid / 3, kk = k - 1 + tid % 3, yy = yb0 - 1 + r; oke = tid < (RI_ROWS + 3) * 3 && yy >= 0 && yy < ih && kk >= 0 && kk < wpr; we = edge[oke ? (size_t)yy * wpr + kk : 0]; }
    }
    const int gxl = gx0 + lx;
    const bool cok = gxl >= 0 && gxl < iw;
#pragma unroll
    for (int i = 0; i < NR; i++) {
      const int r = wv + 4 * i, gy = gy0 + r;
      ok[i] = r < RI_RH && gy >= 0 && gy < ih && cok;
      v[i] = atu(pix, ok[i] ? (unsigned)(gy * iw + gxl) : 0u);
    }
#pragma unroll
    for (int j = 0; j < NS; j++) {
      const int t = tid + 256 * j, r = t >> 4, c = 64 + (t & 15);
      const int gx = gx0 + c, gy = gy0 + r;
      ok[NR + j] = r < RI_RH && c < RI_RW && gx < iw && gy >= 0 && gy < ih;      // (gx >= 54: never negative)
      v[NR + j] = atu(pix, ok[NR + j] ? (unsigned)(gy * iw + gx) : 0u);
    }
#pragma unroll
    for (int i = 0; i < NR; i++) {
      const int r = wv + 4 * i;
      if (r < RI_RH) { col[r * RI_RW + lx] = v[i]; lnk[r * RI_RW + lx] = ok[i] ? 0 : -1; }
    }
#pragma unroll
    for (int j = 0; j < NS; j++) {
      const int t = tid + 256 * j, r = t >> 4, c = 64 + (t & 15);
      if (r < RI_RH && c < RI_RW) { col[r * RI_RW + c] = v[NR + j]; lnk[r * RI_RW + c] = ok[NR + j] ? 0 : -1; }
    }
    if (tid < (RI_ROWS + 2) * 2) sm[tid] = okm ? wm : 0ull;
    if (tid < (RI_ROWS + 3) * 3) se[tid] = oke ? we : 0ull;
  }
  constexpr int MW = 64 + 2, MH = RI_ROWS + 2, MN = MW * MH;       // cells (RI_H .. RI_H + 65, RI_H .. RI_H + RI_ROWS + 1)
  __syncthreads();
  (void)MN;
  {
    auto cell = [&](int cy, int cx) {
      const int gx = gx0 + RI_H + cx, gy = gy0 + RI_H + cy;
      const unsigned mbit = (unsigned)(sm[cy * 2 + (cx >> 6)] >> (cx & 63)) & 1u, ebit = (unsigned)(se[(cy + 1) * 3 + 1 + (cx >> 6)] >> (cx & 63)) & 1u;
      alw[(RI_H + cy) * RI_RW + RI_H + cx] = (unsigned char)((gx < iw && gy < ih) ? (mbit | (ebit ? 0u : 2u)) : 0u);      // bit 0: mask set, bit 1: NOT a strong edge
    };
    for (int cy = rd_ty(); cy < MH; cy += 4) cell(cy, (int)threadIdx.x);            // (rows by waves, columns 0..63 by lanes: no division)
    if (tid < 2 * MH) cell(tid >> 1, 64 + (tid & 1));                              // columns 64, 65
    static_assert(MW == 66, "two columns beyond the lanes");
  }
  // raw links (cells of the first row / column of the region cannot know theirs: nothing reads them, see RI_H)
  {
    auto cell = [&](int cy, int cx) {
      const int t = cy * RI_RW + cx;
      if (lnk[t] < 0) return;
      const int gx = gx0 + cx, gy = gy0 + cy;
      int l = t;
      if (gy > 0 && cy > 0 && col[t] == col[t - RI_RW]) l = t - RI_RW;
      else if (gx > 0 && cx > 0 && col[t] == col[t - 1]) l = t - 1;
      lnk[t] = (short)l;
    };
    for (int cy = rd_ty(); cy < RI_RH; cy += 4) cell(cy, (int)threadIdx.x);
    for (int t = tid; t < RI_RH * 16; t += 256) { const int c = 64 + (t & 15); if (c < RI_RW) cell(t >> 4, c); }
  }
  __syncthreads();
  // allowed directions for the cells whose proposals are needed: the tile and one more row / column (in place of the two bits: a cell's
  // byte is read by itself - its own bits - and by its left and upper neighbours - bit 1 -, so the bytes are rewritten after a barrier)
  // (cells of the 65 x (RI_ROWS + 1) block: rows by waves - slot i = row wv + 4 i -, columns 0..63 by lanes, column 64 a cell of the first RI_ROWS + 1 threads: no division)
  constexpr int NPR = (RI_ROWS + 1 + 3) / 4, NP = NPR + 1;
  const int wvp = rd_ty(), lxp = threadIdx.x;
  int pcx[NP], pcy[NP];
  bool pin[NP];
#pragma unroll
  for (int i = 0; i < NP; i++) {
    pcy[i] = i < NPR ? wvp + 4 * i : tid;
    pcx[i] = i < NPR ? lxp : 64;
    pin[i] = pcy[i] < RI_ROWS + 1;
  }
  unsigned char anew[NP];
#pragma unroll
  for (int i = 0; i < NP; i++) {
    unsigned a = 0;
    if (pin[i]) {
      const int cx = RI_H + pcx[i], cy = RI_H + pcy[i], c = cy * RI_RW + cx;
      const int gx = gx0 + cx, gy = gy0 + cy;
      if (gx > 0 && gy > 0 && gx < iw - 1 && gy < ih - 1) {
        const unsigned b0 = alw[c];
        const bool any = (b0 & 1) != 0, z0 = (b0 & 2) != 0;
        const int v = col[c];
        if ((v == col[c - RI_RW] || any) && z0) a |= 1;
        if ((v == col[c - 1] || any) && z0) a |= 2;
        if ((v == col[c + 1] || any) && (alw[c + 1] & 2)) a |= 4;
        if ((v == col[c + RI_RW] || any) && (alw[c + RI_RW] & 2)) a |= 8;
        a |= 16;
      }
    }
    anew[i] = (unsigned char)a;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NP; i++)
    if (pin[i]) alw[(RI_H + pcy[i]) * RI_RW + RI_H + pcx[i]] = anew[i];
  __syncthreads();
  // proposals (the jumps level by level for all of a thread's cells: eight dependent LDS reads per thread, not per cell)
  {
    int m[NP], og[NP], cc[NP];
    bool act[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) {
      const bool in = pin[i];
      const int c = in ? (RI_H + pcy[i]) * RI_RW + RI_H + pcx[i] : RI_H * RI_RW + RI_H;
      const unsigned a = in ? alw[c] : 0u;
      cc[i] = in ? c : -1;
      act[i] = (a & 16) != 0;
      og[i] = act[i] ? lnk[c] : RI_H * RI_RW + RI_H;      // (a cell that proposes nothing walks from the tile's first pixel: always inside the frame)
      int mm = og[i];      // (cell indices order like pixel indices: both are row-major over the same pixels)
      if (a & 1) { const int s = lnk[c - RI_RW]; mm = s < mm ? s : mm; }
      if (a & 2) { const int s = lnk[c - 1]; mm = s < mm ? s : mm; }
      if (a & 4) { const int s = lnk[c + 1]; mm = s < mm ? s : mm; }
      if (a & 8) { const int s = lnk[c + RI_RW]; mm = s < mm ? s : mm; }
      m[i] = mm;
    }
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
      for (int i = 0; i < NP; i++) m[i] = lnk[m[i]];
#pragma unroll
    for (int i = 0; i < NP; i++) if (cc[i] >= 0) prop[cc[i]] = (short)((act[i] && m[i] != og[i]) ? m[i] : 0x7fff);
  }
  __syncthreads();
  // the labels after the first launch, the allow bytes, the sizes' start values
  const int x = rd_b.x * 64 + threadIdx.x;
  if (x >= iw) return;
  for (int r = rd_ty(); r < RI_ROWS; r += 4) {
    const int y = rd_b.y * RI_ROWS + r;
    if (y >= ih) break;
    const int c = (r + RI_H) * RI_RW + threadIdx.x + RI_H, p = y * iw + x;
    int l = lnk[c];
    const int g0 = prop[c];
    l = g0 < l ? g0 : l;
    if (x + 1 < iw && lnk[c + 1] == c) { const int g1 = prop[c + 1]; l = g1 < l ? g1 : l; }
    if (y + 1 < ih && lnk[c + RI_RW] == c) { const int g2 = prop[c + RI_RW]; l = g2 < l ? g2 : l; }
    const int w = ((gy0 + l / RI_RW) * iw + gx0 + l % RI_RW) << 3;
    A[p] = w;
    B[p] = w;
    if (size_out) {
      // rc:74-95: on-pixels of the 3x3 block, the pixel's own included; 1 -> 0; the frame's border and off-pixels 0
      int j = 0;
      const int b = threadIdx.x;
      if (x > 0 && y > 0 && x < iw - 1 && y < ih - 1 && ((se[(r + 1) * 3 + 1] >> b) & 1ull)) {
        int cnt = 0;
#pragma unroll
        for (int dr = 0; dr < 3; dr++) {
          const u64 *w = &se[(r + dr) * 3];
          const unsigned v = b == 0 ? (((unsigned)w[1] & 3u) << 1) | (unsigned)(w[0] >> 63) : (b == 63 ? ((unsigned)(w[1] >> 62) & 3u) | (((unsigned)w[2] & 1u) << 2) : (unsigned)(w[1] >> (b - 1)) & 7u);
          cnt += __popc(v);
        }
        j = cnt == 1 ? 0 : cnt;
      }
      size_out[p] = j;
    }
  }
  // the allow bytes of 8 rows as one word per column (word (y >> 3) * iw + x, byte y & 7): a thread of the rounds takes 8 rows of a column and reads them with one load
  static_assert(RI_ROWS == 32, "four waves, eight rows each");
  {
    const int r0 = rd_ty() * 8, y0 = rd_b.y * RI_ROWS + r0;
    if (y0 < ih) {
      u64 w = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) w |= (u64)((y0 + k < ih) ? alw[(r0 + k + RI_H) * RI_RW + threadIdx.x + RI_H] : 0) << (8 * k);
      allow[(size_t)(y0 >> 3) * iw + x] = w;
    }
  }
}

// rc:300-334, every launch with its work-items running CONCURRENTLY: all of them read the labels the launch began with, propose `min`
// updates for themselves and for their old label's pixel, and the proposals take effect together when the launch ends - a legal
// execution of the reference's kernel (OpenCL promises no work-item that it sees another one's update within a launch), and the one a
// parallel device can reproduce bit for bit.  The reference enqueues the kernel 8 times (rh:325-331), which settles the labels in
// serial orders; with concurrent work-items a 1080p frame needs about 12 launches, so the kernel is launched until a launch changes
// nothing (the CPU restatement the tests compare with is itself checked against the reference's own kernel running that way:
// DESIGN.md, "Region stages").  The rounds start from the reference's raw initial links: the labels a work-item compares are whatever the
// trees hold at that moment, not their roots, and which regions the merge mask joins depends on exactly that.
//
// One launch per round, no launch that applies a round's proposals, and ONE label plane to read per round: the labels live in two
// planes that alternate.  Round r reads every label from plane X (complete: the labels after round r-1) and writes into plane Y, which
// still holds the labels of one round earlier.  Labels only ever go down, so what Y holds is an upper bound of what it should hold after
// this round, and every update - a pixel's own new label, the proposals for tree parents - is an atomicMin into Y: no order among them
// matters, nothing needs clearing, and a pixel whose label is the same in X, in Y and after this round writes nothing at all.  Which
// pixels are "the same in Y" is carried by the words themselves: a word is  label << 3 | m,  m = 1 + (round mod 7) when the word was
// written by a change in that round (so the OTHER plane lags behind for this pixel), else 0; the pixel's thread sees the mark of the
// round before its own and brings the other plane up to date.  Marks are never cleared: an old one only matches again 14 rounds later,
// where it causes a harmless write of the same value.  (Every pixel has a thread, the ones on the frame's ring too: they never adopt,
// but they are parents.)
// Both planes start as copies of the initial links (k_region_init), so only changes are ever written.
// Rounds with an even number read A and write B, odd ones read B and write A, and every budget is even: the last round writes A,
// and once a round has changed nothing the planes agree, so the labels are always taken from A (k_region_size strips the marks).
// flags[round] = "this round proposed something" (a proposal always lowers its pixel's own label), which is what the next round
// and the host test.
// (Before: proposals in planes of their own with round tags, every label read as min(label, proposal) - a second plane to read for
//  every label: 8 MB per round, 5 % of a frame's HBM traffic.)
// Memory-latency bound: every thread handles RR_PX pixels below one another - each is the other's vertical neighbour, so a column of six
// costs 6 + 2 + 12 label loads instead of 30 (measured at full rate, same box: 2, 3, 4, 6, 8 pixels: 2056, 2080, 2083 / 2067, 2079, 2062
// frames/s) - and issues all of their label loads before using any, then the first pointer jumps together.
#ifndef RR_PX
#define RR_PX 8
#endif
#ifndef RR_TY
#define RR_TY 4          // thread rows per block.  (Round 5, once the thread's loads travelled together: 8 x 6 pixels 2646-2696 frames/s, 4 x 6 2755-2774, 4 x 8 2727-2734, 2 x 8 2699-2742,
                         //  8 x 4 2669-2722, 8 x 12 2607-2623 - smaller blocks give their wave slots back sooner; before, with the loads one after the other, 2 / 4 / 8 rows: 2094 / 2118 / 2124.)
#endif
#define RR_MBITS 3
#ifndef RR_DPP
#define RR_DPP 1           // the left / right neighbours' words from the neighbouring lanes (0: loaded)
#endif
#ifndef RR_DEEP
#define RR_DEEP 3          // rounds in which the trees are still the chains of the initial links (a pixel's parent is 1, 10, 91 rows above it)
#endif
__device__ __forceinline__ int rr_label(const int *X, unsigned q) { return at32(X, q) >> RR_MBITS; }
// (Block -> tile and frame: rd_block_tile, rd_device.h - a tile's neighbours and the pixels its labels name, the rows above it, are served by the L2 that fetched
//  them for the neighbouring tiles: 64 % of this kernel's L2 requests missed before.)
// PHASE: 1 = launch 1 (proposals meet in the tile), 2 = the other launches that climb raw chains (every pixel hooks its own parent), 3 = the rest (parents combined per block)
template <int PHASE>
__global__ __launch_bounds__(64 * RR_TY) void k_region_round(int *X, int *Y, const u64 *__restrict__ allow, int iw, int ih, int *flags, int round, size_t zs, int gdim) {
  const rd_tile rd_b = rd_block_tile(gdim);
  if (rd_b.x < 0) return;
  const int bx = rd_b.x, by = rd_b.y;
  RD_ZSHIFTZ(rd_b.z, zs, X, Y, allow, flags);
  if (round > 0 && flags[round - 1] == 0) return;
  __shared__ int hk[PHASE == 3 ? 512 : 1], hv[PHASE == 3 ? 512 : 1];
  __shared__ int tmin[PHASE == 1 ? 64 * RR_TY * RR_PX : 1];      // launch 1 only: the smallest proposal for each pixel of the block's tile (see below)
  const int tid = rd_ty() * 64 + threadIdx.x;
  constexpr bool near = PHASE == 1;           // the trees are still the first launch's: a pixel's parent lies ~10 rows above it, mostly inside the tile
  if (PHASE == 3) for (int t = tid; t < 512; t += 64 * RR_TY) { hk[t] = -1; hv[t] = 0x7fffffff; }
  if (near) for (int t = tid; t < 64 * RR_TY * RR_PX; t += 64 * RR_TY) tmin[t] = 0x7fffffff;
  if (PHASE != 2) __syncthreads();
  const int mark = 1 + round % 7, mark_prev = round > 0 ? 1 + (round - 1) % 7 : 8;      // (8: matches nothing - before round 0 no plane lags)
  // (A wave takes 64 pixels of RR_PX rows.  In the launches that climb raw chains the 64 lanes of a row land on ~20 different 64-byte lines per jump - the chains of neighbouring
  //  columns fall out of step wherever a vertical run ends, 7 % of the pixels of the bench stream - and footprints 16 pixels wide and 4 RR_PX rows tall halve that (simulated: 159 -> 74
  //  lines per pixel row in launch 1); measured +0.3 % with blocks of 8 x 6, -1.5 % with 4 x 8: the lines are not what those launches wait for.  profiles/NOTES_r05.md.)
  const int lx = (int)threadIdx.x, ly = rd_ty() * RR_PX;      // column and first row inside the block's 64 x (RR_TY * RR_PX) tile
  const int yb = by * (RR_TY * RR_PX) + ly;      // the thread's RR_PX pixels lie below one another: each is the other's vertical neighbour
  const int x = bx * 64 + lx;
  int p0[RR_PX], og[RR_PX], g[RR_PX], nx[RR_PX], w0[RR_PX];
  unsigned a[RR_PX];
  bool valid[RR_PX], todo[RR_PX];
  {
    // (unsigned element indices: the loads take the plane's base from scalar registers and a 32-bit offset, no 64-bit address arithmetic per access.
    //  Every load of this kernel is UNCONDITIONAL, with the address of a pixel that exists where the value is not wanted: a load under a condition becomes a
    //  branch with a wait for ALL loads in flight behind it - the thread's loads then travel one after the other.)
    int l[RR_PX][5];
    static_assert(RR_PX == 8, "the allow bytes of a thread's 8 rows are one word (k_region_init)");
    const bool colin = x < iw && yb < ih;
    const u64 aw = allow[colin ? (size_t)(yb >> 3) * iw + x : 0];
#pragma unroll
    for (int k = 0; k < RR_PX; k++) {
      const int y = yb + k;
      valid[k] = x < iw && y < ih;
      p0[k] = valid[k] ? y * iw + x : 0;
      a[k] = valid[k] ? (unsigned)(aw >> (8 * k)) & 0xffu : 0u;
      l[k][0] = at32(X, (unsigned)p0[k]);
    }
    // neighbours: above / below inside the thread's own column, or one load each for the first and the last row; left / right from the neighbouring LANES (wave shifts,
    // no memory instruction) - the wave's outermost lanes take theirs from one more load per row in which the lower half of the wave asks for the pixel left of the tile
    // and the upper half for the pixel right of it (two addresses per load).  Addresses clamped into the plane; whether a neighbour counts depends on the allow bits,
    // which are 0 towards anything outside the frame.
    l[0][1] = at32(X, (unsigned)((valid[0] && yb > 0) ? p0[0] - iw : p0[0]));
    l[RR_PX - 1][4] = at32(X, (unsigned)((valid[RR_PX - 1] && yb + RR_PX - 1 < ih - 1) ? p0[RR_PX - 1] + iw : p0[RR_PX - 1]));
#if RR_DPP
    {
      const int x0 = bx * 64;
      const bool lowhalf = threadIdx.x < 32;
      const int xe = lowhalf ? x0 - 1 : x0 + 64;
      int ed[RR_PX];
#pragma unroll
      for (int k = 0; k < RR_PX; k++) {
        const int y = yb + k;
        ed[k] = at32(X, (unsigned)((y < ih && xe >= 0 && xe < iw) ? y * iw + xe : p0[k]));
      }
#pragma unroll
      for (int k = 0; k < RR_PX; k++) {
        l[k][2] = __builtin_amdgcn_update_dpp(ed[k], l[k][0], 0x138, 0xf, 0xf, false);      // wave_shr:1 - lane i takes lane i - 1's word, lane 0 keeps `ed` (the pixel left of the tile)
        l[k][3] = __builtin_amdgcn_update_dpp(ed[k], l[k][0], 0x130, 0xf, 0xf, false);      // wave_shl:1 - lane i takes lane i + 1's word, lane 63 keeps `ed` (the pixel right of the tile)
      }
    }
#else
#pragma unroll
    for (int k = 0; k < RR_PX; k++) {
      l[k][2] = at32(X, (unsigned)((valid[k] && x > 0) ? p0[k] - 1 : p0[k]));
      l[k][3] = at32(X, (unsigned)((valid[k] && x < iw - 1) ? p0[k] + 1 : p0[k]));
    }
#endif
#pragma unroll
    for (int k = 0; k < RR_PX; k++) {       // (the neighbours inside the thread's own column: no loads; an invalid pixel's word is never used - its neighbour above is on the ring)
      if (k > 0) l[k][1] = l[k - 1][0];
      if (k < RR_PX - 1) l[k][4] = l[k + 1][0];
    }
#pragma unroll
    for (int k = 0; k < RR_PX; k++) {
      int e[5];
      w0[k] = l[k][0];
#pragma unroll
      for (int c = 0; c < 5; c++) e[c] = l[k][c] >> RR_MBITS;
      og[k] = e[0];
      // the smallest label among the pixel itself and the neighbours it may adopt from: a neighbour that is not allowed counts as
      // +infinity (bit k of `a` spread over a word selects the label or 0x7fffffff: two operations per neighbour, no compares)
      int m = e[0];
#pragma unroll
      for (int c = 1; c < 5; c++) {
        const int t = __builtin_amdgcn_sbfe((int)a[k], c - 1, 1);      // -1 if allowed, else 0
        const int xk = (e[c] & t) | (0x7fffffff & ~t);
        m = xk < m ? xk : m;
      }
      g[k] = (a[k] & 16) ? m : e[0];
    }
  }
  // rc:328: the eight pointer jumps (a root maps to itself), level by level for the thread's RR_PX pixels together: seven dependent
  // trips to memory per thread, not seven per pixel - in the first rounds the trees are the raw chains of rc:289-298 and every jump is real
  const float inv_iw = 1.0f / (float)iw;
#pragma unroll
  for (int k = 0; k < RR_PX; k++) {
    const int v = rr_label(X, (unsigned)g[k]);      // (g is a pixel index for every lane: a pixel that does not act holds its own label)
    nx[k] = (a[k] & 16) ? v : g[k];
  }
  for (int j = 1; j < 8; j++) {
    bool moving = false;
#pragma unroll
    for (int k = 0; k < RR_PX; k++) moving = moving || nx[k] != g[k];
    if (!__any(moving)) break;
#pragma unroll
    for (int k = 0; k < RR_PX; k++) { g[k] = nx[k]; nx[k] = rr_label(X, (unsigned)nx[k]); }      // (a pixel that has arrived re-reads its root: same value)
  }
  bool any_todo = false;
#pragma unroll
  for (int k = 0; k < RR_PX; k++) {
    g[k] = (a[k] & 16) ? nx[k] : og[k];          // (after the loop nx is one jump ahead of g, or equal to it; ring pixels keep their label)
    todo[k] = (a[k] & 16) && g[k] != og[k];
    // the pixel's own word in Y: its new label (marked), or - where Y lags behind - the label it keeps
    const bool lag = (w0[k] & 7) == mark_prev;
    if (!near && valid[k] && (todo[k] || lag)) atomicMin(&Y[p0[k]], (g[k] << RR_MBITS) | (todo[k] ? mark : 0));     // (no value comes back: the thread does not wait for it; g == og unless todo)
    any_todo = any_todo || todo[k];
  }
  if (near) {
    // Launch 1: every pixel has a parent of its own and proposes to it - two million atomics on top of the two million own words.  The
    // parent is the pixel ~10 rows above, for four out of five pixels inside the block's tile: those proposals meet in LDS and leave with the
    // parent's own word, as ONE atomic per pixel; the rest go to memory as in the other launches.
    const int origin = by * (RR_TY * RR_PX) * iw + bx * 64;
    bool out[RR_PX];       // proposals for parents outside the tile
#pragma unroll
    for (int k = 0; k < RR_PX; k++) {
      const int d = og[k] - origin;
      bool inside = false;
      int cell = 0;
      if (d >= 0 && d < RR_TY * RR_PX * iw) {        // (below 2^24 for every frame this library accepts rows of: exact in single precision up to the correction)
        int row = (int)((float)d * inv_iw), col = d - row * iw;
        if (col < 0) { row--; col += iw; } else if (col >= iw) { row++; col -= iw; }
        inside = col < 64 && row < RR_TY * RR_PX;
        cell = row * 64 + col;
      }
      if (todo[k] && inside) atomicMin(&tmin[cell], g[k]);
      out[k] = todo[k] && !inside;
    }
    // (the proposals for parents outside the tile - a fifth of the pixels - go without a guarding load: with the guards, read together or pixel by pixel, the launch was no faster)
#pragma unroll
    for (int k = 0; k < RR_PX; k++) { const int w = (g[k] << RR_MBITS) | mark; if (out[k]) atomicMin(&Y[og[k]], w); }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RR_PX; k++) {
      if (!valid[k]) continue;
      const bool lag = (w0[k] & 7) == mark_prev;
      int w = (todo[k] || lag) ? ((g[k] << RR_MBITS) | (todo[k] ? mark : 0)) : 0x7fffffff;
      const int h = tmin[(ly + k) * 64 + lx];
      if (h != 0x7fffffff) { const int wh = (h << RR_MBITS) | mark; w = wh < w ? wh : w; }
      if (w != 0x7fffffff) atomicMin(&Y[p0[k]], w);
    }
    if (__any(any_todo) && threadIdx.x == 0) flags[round] = 1;
    return;
  }
  // Hooking the old parent: once the trees are shallow, all pixels of a tree share one parent, so the block first reduces its
  // (parent -> smallest proposal) pairs in a small LDS hash and then issues one guarded atomic per distinct parent.
  auto hook = [&](int key, int val) {
    const int w = (val << RR_MBITS) | mark;
    if (w < ld_agent(&Y[key])) atomicMin(&Y[key], w);
  };
  bool need[RR_PX];
#pragma unroll
  for (int k = 0; k < RR_PX; k++) {
    // a lane whose (parent, proposal) pair repeats its left neighbour's adds nothing to a min: skip it (most lanes inside a region)
    const int pog = __shfl_up(og[k], 1), pg = __shfl_up(g[k], 1), pt = __shfl_up((int)todo[k], 1);
    need[k] = todo[k] && !(threadIdx.x > 0 && pt && pog == og[k] && pg == g[k]);
  }
  if (PHASE == 2) {       // the first rounds climb the raw chains: every pixel has a parent of its own (the pixel above it), nothing to combine
    // (guard and atomic pixel by pixel: a guard only saves its atomic when the parent's own thread has written already, and the later it looks the more often that is - the
    //  eight guards read together made this launch 195 instead of 151 us; no guards at all cost 3 % of the frame rate: the atomics are what is expensive)
#pragma unroll
    for (int k = 0; k < RR_PX; k++) { const int w = (g[k] << RR_MBITS) | mark; if (need[k] && w < ld_agent(&Y[og[k]])) atomicMin(&Y[og[k]], w); }
  } else {
#pragma unroll
    for (int k = 0; k < RR_PX; k++) {
      if (!need[k]) continue;
      unsigned h = ((unsigned)og[k] * 2654435761u) >> 23;
      int probes = 0;
      for (;;) {
        const int kprev = atomicCAS(&hk[h], -1, og[k]);
        if (kprev == -1 || kprev == og[k]) { atomicMin(&hv[h], g[k]); break; }
        h = (h + 1) & 511;
        if (++probes == 16) { hook(og[k], g[k]); break; }
      }
    }
  }
  if (__any(any_todo) && threadIdx.x == 0) flags[round] = 1;
  if (PHASE != 3) return;
  __syncthreads();
  for (int t = tid; t < 512; t += 64 * RR_TY) {
    const int key = hk[t];
    if (key != -1) hook(key, hv[t]);
  }
}

// rc:336-346: out[label]++ for every pixel.  Most pixels belong to a handful of huge regions, so counts are
// aggregated per wave (runs of equal labels among its 64 consecutive pixels), then per block in an LDS hash, before touching
// global atomics.
#define RS_T 1024
#ifndef RS_PER_THREAD
#define RS_PER_THREAD 32
#endif
__device__ __forceinline__ void rs_accum(int *keys, int *vals, int *out, int label, int cnt) {
  unsigned h = ((unsigned)label * 2654435761u) >> 22;
  int probes = 0;
  for (;;) {
    const int kprev = atomicCAS(&keys[h], -1, label);
    if (kprev == -1 || kprev == label) { atomicAdd(&vals[h], cnt); return; }
    h = (h + 1) & (RS_T - 1);
    if (++probes == 32) { atomicAdd(&out[label], cnt); return; }
  }
}

// (marked: the plane holds the words of k_region_round - label << 3 | mark; the plain labels are stored back on the way)
__global__ __launch_bounds__(256) void k_region_size(int *out, int *__restrict__ label, int n, int *zero_me, int marked, size_t zs) {
  RD_ZSHIFT(zs, out, label, zero_me);
  if (zero_me && blockIdx.x == 0 && threadIdx.x == 0) *zero_me = 0;     // (a counter of the next stage: saves a fill launch)
  __shared__ int keys[RS_T], vals[RS_T];
  for (int i = threadIdx.x; i < RS_T; i += 256) { keys[i] = -1; vals[i] = 0; }
  __syncthreads();
  const int begin = blockIdx.x * 256 * RS_PER_THREAD;
  int lks[RS_PER_THREAD];
#pragma unroll
  for (int k = 0; k < RS_PER_THREAD; k++) {     // all loads first: one block per CU would otherwise wait for them one by one
    const int i = begin + k * 256 + threadIdx.x;
    lks[k] = i < n ? label[i] : -1;
  }
  if (marked) {
#pragma unroll
    for (int k = 0; k < RS_PER_THREAD; k++) {
      const int i = begin + k * 256 + threadIdx.x;
      if (i < n) { lks[k] >>= RR_MBITS; label[i] = lks[k]; }
    }
  }
#pragma unroll
  for (int k = 0; k < RS_PER_THREAD; k++) {
    const int lk = lks[k];
    const int l0 = __shfl(lk, 0);
    if (__all(lk == l0)) {          // the usual case: the whole wave lies inside one region
      if ((threadIdx.x & 63) == 0 && l0 != -1) rs_accum(keys, vals, out, l0, 64);
      continue;
    }
    // the lanes are consecutive pixels: every run of equal labels is counted by its first lane
    const int lane = threadIdx.x & 63;
    const int prev = __shfl_up(lk, 1);
    const bool start = lane == 0 || prev != lk;
    const unsigned long long after = __ballot(start) & ~((2ull << lane) - 1ull);
    if (start && lk != -1) rs_accum(keys, vals, out, lk, (after ? __ffsll((long long)after) - 1 : 64) - lane);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < RS_T; i += 256)
    if (keys[i] != -1 && vals[i] != 0) atomicAdd(&out[keys[i]], vals[i]);
}

// rc:348-371.  The reference updates labels in place in raster order: a small-region pixel sees the NEW labels of
// its NW, N, NE, W neighbours.  One Jacobi round of that recurrence: `cur` holds the previous round's
// values for small-region pixels; the rounds are iterated by the launcher.
// (all loads of a pick are issued before any is used: the loops are unrolled and the frame-border tests only mask the result)
__device__ __forceinline__ int despeckle2_pick(const int *__restrict__ cur, const int *__restrict__ old, const int *__restrict__ size, int l0, int x, int y, int iw, int ih) {
  const int p0 = y * iw + x;
  int lab[9], sz[9];
  bool ok[9];
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const int xx = k % 3 - 1, yy = k / 3 - 1;
    ok[k] = !(x + xx < 0 || x + xx >= iw || y + yy < 0 || y + yy >= ih);
    const int p1 = ok[k] ? p0 + yy * iw + xx : p0;
    lab[k] = k < 4 ? cur[p1] : old[p1];          // NW, N, NE, W come before p in raster order
  }
#pragma unroll
  for (int k = 0; k < 9; k++) sz[k] = size[lab[k]];
  int res = l0, maxSize = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) if (ok[k] && sz[k] > maxSize) { maxSize = sz[k]; res = lab[k]; }
  return res;
}

// first round over the whole plane (cur == old); the pixels of small regions - the only ones later rounds can change -
// are appended to `list` (any order), *count = their number.  One block per 64x32 tile collects its pixels in LDS and
// reserves its share of the list with a single atomic (same-address atomics cost ~8 ns each: one per wave was 250 us).
#define D2_ROWS 32
__global__ __launch_bounds__(256) void k_despeckle2_first(int *__restrict__ nxt, int *__restrict__ other, int *__restrict__ stamp, int *__restrict__ list, int *count,
                                                           const int *__restrict__ old, const int *__restrict__ size, int thre, int iw, int ih) {
  __shared__ int loc[64 * D2_ROWS];
  __shared__ int nloc, base;
  if (threadIdx.x == 0 && rd_ty() == 0) {
    nloc = 0;
    if (blockIdx.x == 0 && blockIdx.y == 0) count[1] = count[2] = 0;     // the counters of the two other work lists
  }
  __syncthreads();
  const int x = blockIdx.x * 64 + threadIdx.x;
  // the labels and region sizes of this thread's D2_ROWS / 4 pixels: two levels of loads, each issued for all of them together
  int l0s[D2_ROWS / 4], szs[D2_ROWS / 4];
#pragma unroll
  for (int k = 0; k < D2_ROWS / 4; k++) {
    const int y = blockIdx.y * D2_ROWS + rd_ty() + 4 * k;
    l0s[k] = old[(x < iw && y < ih) ? y * iw + x : 0];
  }
#pragma unroll
  for (int k = 0; k < D2_ROWS / 4; k++) szs[k] = size[l0s[k]];
#pragma unroll
  for (int k = 0; k < D2_ROWS / 4; k++) {
    const int r = rd_ty() + 4 * k;
    const int y = blockIdx.y * D2_ROWS + r;
    const bool inside = x < iw && y < ih;
    const int p0 = y * iw + x;
    bool small = false;
    if (inside) {
      const int l0 = l0s[k];
      small = szs[k] <= thre;
      nxt[p0] = small ? despeckle2_pick(old, old, size, l0, x, y, iw, ih) : l0;
      other[p0] = l0;
      stamp[p0] = small ? 0 : 0x7fffffff;      // work-list stamps (k_despeckle2_active): other pixels never enter a list
    }
    const unsigned long long m = __ballot(small);
    if (m) {
      const int lane = threadIdx.x, leader = __ffsll((long long)m) - 1;
      int b = 0;
      if (lane == leader) b = atomicAdd(&nloc, __popcll(m));
      b = __shfl(b, leader);
      if (small) loc[b + __popcll(m & ((1ull << lane) - 1))] = p0;
    }
  }
  __syncthreads();
  const int tid = rd_ty() * 64 + threadIdx.x;
  if (tid == 0 && nloc > 0) base = atomicAdd(count, nloc);
  __syncthreads();
  for (int i = tid; i < nloc; i += 256) list[base + i] = loc[i];
}

// TWO Jacobi rounds per launch, exactly: the value of a pixel after round r+2 needs the round r+1 values of its four earlier
// neighbours, which are recomputed here from the round r plane (a launch costs more than the extra gathers: the rounds are
// latency-bound at a few microseconds each, and the recurrence needs a few dozen of them - see despeckle2()).
//
// Only the pixels that can still move are visited.  `cur` holds round r everywhere, `nxt` holds round r-2 and differs from `cur`
// exactly on the pixels the previous launch changed.  A pixel's value after round r+2 is a function of the round r values of
// its two-step earlier neighbourhood, so a pixel keeps its value unless the previous launch changed something in there; the
// work list of a launch is therefore: the pixels the previous launch changed (their `nxt` is stale) and every small-region
// pixel up to two steps later in raster order than one of those (offsets D2_LATER).  Each launch builds the next list while it
// runs: a pixel goes in once (`stamp` carries the launch number), through a block-local list and one reservation per block.
#define D2A_CAP 6144
__constant__ signed char D2_LATER[11][2] = { {1, 0}, {2, 0}, {-1, 1}, {0, 1}, {1, 1}, {2, 1}, {-2, 2}, {-1, 2}, {0, 2}, {1, 2}, {2, 2} };
__device__ __forceinline__ void d2_flush(int *__restrict__ list_out, int *count_out, const int *loc, int &nloc, int &base, int tid) {
  __syncthreads();
  const int m = min(nloc, D2A_CAP);
  if (tid == 0 && m > 0) base = atomicAdd(count_out, m);
  __syncthreads();
  for (int i = tid; i < m; i += 256) list_out[base + i] = loc[i];
  __syncthreads();
  if (tid == 0) nloc = 0;
  __syncthreads();
}
// (blk of nblk blocks of 256 threads; loc / nloc / base: the block's LDS)
__device__ __forceinline__ void d2_active_rounds(int *__restrict__ nxt, const int *__restrict__ cur, const int *__restrict__ list, const int *__restrict__ count,
                                                 int *__restrict__ list_out, int *count_out, int *count_zero, int *work_trace, int *__restrict__ stamp, int tag,
                                                 const int *__restrict__ old, const int *__restrict__ size, int thre, int iw, int ih, int blk, int nblk, int *loc, int &nloc, int &base) {
  const int tid = threadIdx.x;
  if (tid == 0) {
    nloc = 0;
    if (blk == 0) *count_zero = 0;        // the list after next: nobody reads or appends to it during this launch
  }
  __syncthreads();
  const int n = *count;
  if (blk == 0 && tid == 0) *work_trace = n;   // (diagnostic: the work list length of each launch)
  for (int j0 = blk * 256; j0 < n; j0 += nblk * 256) {
    const int j = j0 + tid;
    bool changed = false;
    int p0 = 0, x = 0, y = 0;
    if (j < n) {
      p0 = list[j];
      x = p0 % iw, y = p0 / iw;
      // The launches are latency-bound (a few hundred pixels, every dependent gather a trip to L2), so the gathers are arranged
      // in two levels only: the round-r and input labels of the window (x-2..x+2, y-2..y+1) - the cells that are not needed
      // fall away at compile time - then the region sizes of all of them; the rest is register arithmetic.
      int C[4][5], O[4][5], CS[4][5], OS[4][5];
      bool in[4][5];
#pragma unroll
      for (int wy = 0; wy < 4; wy++)
#pragma unroll
        for (int wx = 0; wx < 5; wx++) {
          const int xx = x + wx - 2, yy = y + wy - 2;
          in[wy][wx] = xx >= 0 && xx < iw && yy >= 0 && yy < ih;
          const int q = in[wy][wx] ? p0 + (wy - 2) * iw + (wx - 2) : p0;
          C[wy][wx] = cur[q];
          O[wy][wx] = old[q];
        }
#pragma unroll
      for (int wy = 0; wy < 4; wy++)
#pragma unroll
        for (int wx = 0; wx < 5; wx++) { CS[wy][wx] = size[C[wy][wx]]; OS[wy][wx] = size[O[wy][wx]]; }
      int lab[9], sz[9];
      bool ok[9];
#pragma unroll
      for (int k = 0; k < 9; k++) { ok[k] = in[k / 3 + 1][k % 3 + 1]; lab[k] = O[k / 3 + 1][k % 3 + 1]; sz[k] = OS[k / 3 + 1][k % 3 + 1]; }
#pragma unroll
      for (int k = 0; k < 4; k++)      // an earlier neighbour that belongs to a small region: its value after round r+1 (despeckle2_pick on the window)
        if (ok[k] && sz[k] <= thre) {
          int r = lab[k], rs = sz[k], m = 0;
#pragma unroll
          for (int i = 0; i < 9; i++) {
            const int wy = k / 3 + i / 3, wx = k % 3 + i % 3;
            const int l = i < 4 ? C[wy][wx] : O[wy][wx], sl = i < 4 ? CS[wy][wx] : OS[wy][wx];
            if (in[wy][wx] && sl > m) { m = sl; r = l; rs = sl; }
          }
          lab[k] = r; sz[k] = rs;
        }
      int res = lab[4], maxSize = 0;
#pragma unroll
      for (int k = 0; k < 9; k++) if (ok[k] && sz[k] > maxSize) { maxSize = sz[k]; res = lab[k]; }
      nxt[p0] = res;
      changed = res != C[2][2];
    }
    if (list_out != nullptr) {
      if (changed) {
        int q[12], st[12];
        q[11] = p0;
#pragma unroll
        for (int k = 0; k < 11; k++) {
          const int xx = x + D2_LATER[k][0], yy = y + D2_LATER[k][1];
          q[k] = (xx >= 0 && xx < iw && yy < ih) ? yy * iw + xx : p0;
        }
        // stamps: 0x7fffffff on pixels of large regions (stay out), otherwise the number of the last launch that listed the pixel
#pragma unroll
        for (int k = 0; k < 12; k++) st[k] = atomicMax(&stamp[q[k]], tag);
#pragma unroll
        for (int k = 0; k < 12; k++)
          if (st[k] < tag) {
            const int i = atomicAdd(&nloc, 1);
            if (i < D2A_CAP) loc[i] = q[k]; else list_out[atomicAdd(count_out, 1)] = q[k];     // (never: at most 256 x 12 per pass, flushed below)
          }
      }
      __syncthreads();
      if (nloc + 256 * 12 > D2A_CAP) d2_flush(list_out, count_out, loc, nloc, base, tid);
    }
  }
  if (list_out != nullptr) d2_flush(list_out, count_out, loc, nloc, base, tid);
}

__global__ __launch_bounds__(256) void k_despeckle2_active(int *__restrict__ nxt, const int *__restrict__ cur, const int *__restrict__ list, const int *__restrict__ count,
                                                            int *__restrict__ list_out, int *count_out, int *count_zero, int *work_trace, int *__restrict__ stamp, int tag,
                                                            const int *__restrict__ old, const int *__restrict__ size, int thre, int iw, int ih) {
  __shared__ int loc[D2A_CAP];
  __shared__ int nloc, base;
  d2_active_rounds(nxt, cur, list, count, list_out, count_out, count_zero, work_trace, stamp, tag, old, size, thre, iw, ih, blockIdx.x, gridDim.x, loc, nloc, base);
}

// ---- the absorption, exactly (fast path) -------------------------------------------------------------------------------------------
// rc:348-371 in the reference's serial raster order IS a recurrence on a DAG: a pixel of a small region takes the label of the
// largest region among  new(NW), new(N), new(NE), new(W), old(C), old(E), old(SW), old(S), old(SE)  (first maximum in that
// order), every other pixel keeps its label.  Two facts make it cheap to evaluate exactly:
//   (1) chains of dependent small-region pixels are short almost everywhere (on the bench stream 97 % of the 74 k small-region
//       pixels of a frame hang on chains of at most 8 pixels), so a tile that also loads a halo of AB_HK pixels to its left, top
//       and right can evaluate nearly all of its own pixels from LDS, and it KNOWS which ones: a value is exact once the values
//       of the pixel's small-region predecessors are exact, small-region pixels on the rim of what the tile loaded never are;
//   (2) for the rest (a few runs along the frame's last row, as a rule: ~2 k pixels per 1080p frame)  T(p) = size of new(p)
//       obeys  T(p) = max(T(NW), T(N), T(NE), T(W), sizes of the five old labels),  a pure max-propagation, and given T the label
//       is that of the first predecessor with the same T (a pointer) or the first old label with that size (a constant):
//       pointer doubling along the W links settles a run of a thousand pixels in ten steps instead of a thousand.
// k_absorb_tile does (1) and lists the pixels it cannot decide (their `out` word = -(list index) - 2), k_absorb_tail does (2)
// for up to AT_CAP of them in one block.  Whatever they cannot finish (status[0] != 0) is left to despeckle2_slow().
#define AB_TW 64
#ifndef AB_TH
#define AB_TH 64        // (64 x 64 tiles: 510 blocks at 1920 x 1080, two per CU - all resident at once - and 1.96 loaded cells per pixel; 64 x 32: 2.39)
#endif
#ifndef AB_HK
#define AB_HK 16
#endif
#define AB_LW (AB_TW + 2 * AB_HK + 2)       // loaded cells: the tile, AB_HK columns left and right, AB_HK rows above, and one more ring for the old 3x3 neighbourhoods
#define AB_LH (AB_TH + AB_HK + 2)
#define AB_NC (AB_LW * AB_LH)
#ifndef AB_NT
#define AB_NT 512
#endif
#define AB_R 4                              // small-region cells per thread (more in a tile: its small-region pixels go to the tail)
#define AB_INEXACT 0x40000000u
typedef unsigned long long ab_word;         // a cell: (size of its label | AB_INEXACT) << 32 | label  (one 64-bit LDS word: readers always see a consistent pair)
__device__ __forceinline__ ab_word ab_pack(int lab, unsigned t) { return ((ab_word)t << 32) | (unsigned)lab; }
__device__ __forceinline__ int ab_lab(ab_word w) { return (int)(unsigned)w; }
__device__ __forceinline__ unsigned ab_t(ab_word w) { return (unsigned)(w >> 32); }

// A record per undecided pixel (AB_REC ints): [0] pixel, [1] best old label, [2] its size, [3 + j] predecessor j (NW, N, NE, W): its final
// label if it is decided (-1: outside the frame) or -(its pixel index) - 2 if it is undecided itself, [7 + j] the size of that label (0: none).
#define AB_REC 12
__device__ __forceinline__ void ab_record(int *__restrict__ rec, int p, int rl, unsigned rs, const ab_word *cell, int c, int iw, int gx0, int gy0) {
  const int d[4] = { -AB_LW - 1, -AB_LW, -AB_LW + 1, -1 };
  int v[AB_REC];
  v[0] = p; v[1] = rl; v[2] = (int)rs; v[11] = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int cj = c + d[j];
    const ab_word w = cell[cj];
    const bool undecided = (ab_t(w) & AB_INEXACT) != 0;
    v[3 + j] = undecided ? -((gy0 + cj / AB_LW) * iw + gx0 + cj % AB_LW) - 2 : ab_lab(w);
    v[7 + j] = undecided ? 0 : (int)ab_t(w);
  }
#pragma unroll
  for (int q = 0; q < AB_REC; q += 4) *(int4 *)(rec + q) = make_int4(v[q], v[q + 1], v[q + 2], v[q + 3]);
}

// (list: room for reccap records)
__global__ __launch_bounds__(AB_NT) void k_absorb_tile(int *__restrict__ out, int *__restrict__ list, int reccap, int *count, const int *__restrict__ old, const int *__restrict__ size,
                                                       int thre, int iw, int ih, size_t zs, int gdim) {
  const rd_tile rd_b = rd_block_tile(gdim);
  if (rd_b.x < 0) return;
  RD_ZSHIFTZ(rd_b.z, zs, out, list, count, old, size);
  __shared__ ab_word cell[AB_NC];
  __shared__ unsigned short slist[AB_NT * AB_R];
  __shared__ unsigned short rlist[AB_TW * AB_TH];      // undecided pixels of the tile (cell indices)
  __shared__ int nsmall, nres, base;
  const int tid = threadIdx.x;
  if (tid == 0) { nsmall = 0; nres = 0; }
  const int gx0 = rd_b.x * AB_TW - AB_HK - 1, gy0 = rd_b.y * AB_TH - AB_HK - 1;
  // old labels and their region sizes for every loaded cell (cells outside the frame: size 0 - they never win a comparison)
  {
    constexpr int IT = (AB_NC + AB_NT - 1) / AB_NT;
    int l[IT], sz[IT];
    bool in[IT];
#pragma unroll
    for (int i = 0; i < IT; i++) {
      const int t = tid + i * AB_NT;
      const int cx = t % AB_LW, cy = t / AB_LW;
      const int gx = gx0 + cx, gy = gy0 + cy;
      in[i] = t < AB_NC && gx >= 0 && gx < iw && gy >= 0 && gy < ih;
      l[i] = old[in[i] ? (unsigned)(gy * iw + gx) : 0u];
    }
#pragma unroll
    for (int i = 0; i < IT; i++) sz[i] = size[(unsigned)l[i]];
    __syncthreads();        // (nsmall = 0 is visible)
#pragma unroll
    for (int i = 0; i < IT; i++) {
      const int t = tid + i * AB_NT;
      if (t >= AB_NC) continue;
      const int cx = t % AB_LW, cy = t / AB_LW;
      const bool small = in[i] && sz[i] <= thre;
      const bool inner = cx >= 1 && cx <= AB_LW - 2 && cy >= 1 && cy <= AB_LH - 2;      // cells this block evaluates (their 3x3 neighbourhood is loaded)
      cell[t] = ab_pack(in[i] ? l[i] : -1, in[i] ? ((unsigned)sz[i] | (small ? AB_INEXACT : 0u)) : 0u);
      if (small && inner) {
        const int k = atomicAdd(&nsmall, 1);
        if (k < AB_NT * AB_R) slist[k] = (unsigned short)t;
      }
    }
  }
  __syncthreads();
  const int ns = nsmall;
  int c[AB_R], rl[AB_R], slot[AB_R];
  unsigned rs[AB_R];
  bool open[AB_R];
#pragma unroll
  for (int k = 0; k < AB_R; k++) { open[k] = false; slot[k] = -1; c[k] = AB_LW + 1; rl[k] = 0; rs[k] = 0; }
  const bool owned = ns > 0 && ns <= AB_NT * AB_R;       // (more small-region cells than the block has registers for: all of them stay undecided)
  if (owned) {
    // the best of the five old labels (C, E, SW, S, SE) per cell: these never change
#pragma unroll
    for (int k = 0; k < AB_R; k++) {
      const int i = tid + k * AB_NT;
      open[k] = i < ns;
      c[k] = open[k] ? slist[i] : AB_LW + 1;
      const ab_word w0 = cell[c[k]], w1 = cell[c[k] + 1], w2 = cell[c[k] + AB_LW - 1], w3 = cell[c[k] + AB_LW], w4 = cell[c[k] + AB_LW + 1];
      rl[k] = ab_lab(w0); rs[k] = ab_t(w0) & ~AB_INEXACT;
      unsigned t;
      t = ab_t(w1) & ~AB_INEXACT; if (t > rs[k]) { rs[k] = t; rl[k] = ab_lab(w1); }
      t = ab_t(w2) & ~AB_INEXACT; if (t > rs[k]) { rs[k] = t; rl[k] = ab_lab(w2); }
      t = ab_t(w3) & ~AB_INEXACT; if (t > rs[k]) { rs[k] = t; rl[k] = ab_lab(w3); }
      t = ab_t(w4) & ~AB_INEXACT; if (t > rs[k]) { rs[k] = t; rl[k] = ab_lab(w4); }
    }
    __syncthreads();
    // rounds: a cell whose four predecessors are exact takes its final value and becomes exact; until nothing moves any more, AB_HK + 8 rounds at most
    for (int round = 0; round < AB_HK + 8; round++) {      // (longer chains - the runs along the frame's outermost rows and columns - are the tail's: it settles them in log steps, here a pixel per round)
      bool progress = false;
#pragma unroll
      for (int k = 0; k < AB_R; k++) {
        if (!open[k]) continue;
        const ab_word p0 = cell[c[k] - AB_LW - 1], p1 = cell[c[k] - AB_LW], p2 = cell[c[k] - AB_LW + 1], p3 = cell[c[k] - 1];
        if ((ab_t(p0) | ab_t(p1) | ab_t(p2) | ab_t(p3)) & AB_INEXACT) continue;
        unsigned bs = 0; int bl = rl[k];
        if (ab_t(p0) > bs) { bs = ab_t(p0); bl = ab_lab(p0); }
        if (ab_t(p1) > bs) { bs = ab_t(p1); bl = ab_lab(p1); }
        if (ab_t(p2) > bs) { bs = ab_t(p2); bl = ab_lab(p2); }
        if (ab_t(p3) > bs) { bs = ab_t(p3); bl = ab_lab(p3); }
        if (rs[k] > bs) { bs = rs[k]; bl = rl[k]; }
        cell[c[k]] = ab_pack(bl, bs);
        open[k] = false;
        progress = true;
      }
      if (!__syncthreads_or(progress)) break;
    }
    // the undecided cells among the tile's own pixels take a slot of the tile's share of the list
#pragma unroll
    for (int k = 0; k < AB_R; k++) {
      const int cx = c[k] % AB_LW - (AB_HK + 1), cy = c[k] / AB_LW - (AB_HK + 1);
      if (open[k] && cx >= 0 && cx < AB_TW && cy >= 0 && cy < AB_TH) slot[k] = atomicAdd(&nres, 1);
    }
  }
  // the tile's own pixels: final labels (the undecided ones are written below)
  const int tx = tid & 63, x = rd_b.x * AB_TW + tx;
  for (int r = tid >> 6; r < AB_TH; r += AB_NT / 64) {
    const int y = rd_b.y * AB_TH + r;
    if (x >= iw || y >= ih) continue;
    const int ci = (r + AB_HK + 1) * AB_LW + tx + AB_HK + 1;
    const ab_word w = cell[ci];
    if (!(ab_t(w) & AB_INEXACT)) out[(unsigned)(y * iw + x)] = ab_lab(w);
    else if (!owned) rlist[atomicAdd(&nres, 1)] = (unsigned short)ci;
  }
  __syncthreads();
  const int nr = nres;
  if (nr == 0) return;
  if (tid == 0) base = atomicAdd(count, nr);
  __syncthreads();
  if (owned) {
#pragma unroll
    for (int k = 0; k < AB_R; k++) {
      if (slot[k] < 0) continue;
      const int p = (gy0 + c[k] / AB_LW) * iw + gx0 + c[k] % AB_LW, at = base + slot[k];
      out[(unsigned)p] = -at - 2;
      if (at < reccap) ab_record(list + (size_t)at * AB_REC, p, rl[k], rs[k], cell, c[k], iw, gx0, gy0);
    }
  } else {
    for (int i = tid; i < nr; i += AB_NT) {       // (no rounds ran: the cells still hold the old labels)
      const int ci = rlist[i], p = (gy0 + ci / AB_LW) * iw + gx0 + ci % AB_LW, at = base + i;
      out[(unsigned)p] = -at - 2;
      if (at >= reccap) continue;
      const ab_word w0 = cell[ci], w1 = cell[ci + 1], w2 = cell[ci + AB_LW - 1], w3 = cell[ci + AB_LW], w4 = cell[ci + AB_LW + 1];
      int l = ab_lab(w0); unsigned m = ab_t(w0) & ~AB_INEXACT, t;
      t = ab_t(w1) & ~AB_INEXACT; if (t > m) { m = t; l = ab_lab(w1); }
      t = ab_t(w2) & ~AB_INEXACT; if (t > m) { m = t; l = ab_lab(w2); }
      t = ab_t(w3) & ~AB_INEXACT; if (t > m) { m = t; l = ab_lab(w3); }
      t = ab_t(w4) & ~AB_INEXACT; if (t > m) { m = t; l = ab_lab(w4); }
      ab_record(list + (size_t)at * AB_REC, p, l, m, cell, ci, iw, gx0, gy0);
    }
  }
}

// The undecided pixels of a frame (n = *count of them, any order) in ONE block; see (2) above.
//   W[i] = (x, lw, ln) in one 64-bit LDS word (readers always see a consistent triple): x = lower bound of T(i) that only ever holds sizes
//   of true ancestors; lw / ln = the undecided pixel that the doubling along the row / along the column has reached (AT_NONE: it has
//   passed the start of the run).  The long chains are runs of small-region pixels along the frame's outermost rows and columns - which
//   the merge never touches (rc:302) - a thousand pixels long: a sweep = doubling steps until no link is left (about eleven), and every
//   step also relaxes the two diagonal predecessors.  Sweeps repeat (links restored) until x is a fixed point of the recurrence.
//   status[0] <- 1 if n > AT_CAP or the sweeps did not settle within AT_SWEEPS (then `out` keeps undecided words), status[1] <- n, status[2] <- sweeps
#define AT_NT 1024
#define AT_CAP 16383
#define AT_NONE 0x3fffu
#define AT_SWEEPS 64
#define AT_XMASK 0x3ffffffull      // 26 bits: region sizes stay below 2^25 + 9 (rd_detector_create limits the frame)
__device__ __forceinline__ ab_word at_pack(unsigned x, unsigned lw, unsigned ln) { return (ab_word)x | ((ab_word)lw << 26) | ((ab_word)ln << 40); }
__device__ __forceinline__ unsigned at_x(ab_word w) { return (unsigned)(w & AT_XMASK); }
__device__ __forceinline__ unsigned at_lw(ab_word w) { return (unsigned)(w >> 26) & AT_NONE; }
__device__ __forceinline__ unsigned at_ln(ab_word w) { return (unsigned)(w >> 40) & AT_NONE; }
__device__ __forceinline__ unsigned at_relax(const ab_word *W, unsigned ref, unsigned m) {
  if (ref != AT_NONE) { const unsigned t = at_x(W[ref]); m = t > m ? t : m; }
  return m;
}

// EPT pixels per thread (compile time: what a thread keeps per pixel between the phases - its four references - stays in registers; the
// record itself is read twice, before and after the sweeps, rather than kept); the kernel picks the smallest EPT that covers n.
template <int EPT>
__device__ __forceinline__ void at_body(ab_word *W, int *__restrict__ out, const int *__restrict__ list, int n, const int *__restrict__ size, int nsteps, int *status) {
  const int tid = threadIdx.x;
  const unsigned long long t0 = wall_clock64();       // (100 MHz: the phases' durations go to status[4..7] in units of 10 ns - diagnostics)
  unsigned rdiag[EPT], rwn[EPT];      // references NW | NE << 16 and W | N << 16 (indices into the list; AT_NONE: not an undecided pixel)
  // the records, then - for predecessors that are undecided themselves - their places in the list (their words in `out`); four pixels at
  // a time (two dependent trips to memory per group: more in flight would not fit the registers of a 1024-thread block)
  constexpr int G = EPT < 4 ? EPT : 4;
#pragma unroll
  for (int k0 = 0; k0 < EPT; k0 += G) {
    int pq[G][4];
    unsigned c0[G];
#pragma unroll
    for (int kk = 0; kk < G; kk++) {
      const int i = k0 + kk < EPT ? tid + (k0 + kk) * AT_NT : n;
      const int4 *rec = (const int4 *)(list + (size_t)(i < n ? i : 0) * AB_REC);
      const int4 a = rec[0], b = rec[1], d = rec[2];
      pq[kk][0] = a.w; pq[kk][1] = b.x; pq[kk][2] = b.y; pq[kk][3] = b.z;
      unsigned c = (unsigned)a.z;
      c = (unsigned)b.w > c ? (unsigned)b.w : c; c = (unsigned)d.x > c ? (unsigned)d.x : c; c = (unsigned)d.y > c ? (unsigned)d.y : c; c = (unsigned)d.z > c ? (unsigned)d.z : c;
      c0[kk] = c;
    }
    // (a predecessor that this pixel's tile could not decide - it lies in the tile's halo - may have been decided by its own tile: then
    //  its word in `out` is a label, and its size joins the constant part)
    int pv[G][4];
#pragma unroll
    for (int kk = 0; kk < G; kk++) {
      const int i = k0 + kk < EPT ? tid + (k0 + kk) * AT_NT : n;
#pragma unroll
      for (int j = 0; j < 4; j++) pv[kk][j] = (i < n && pq[kk][j] < -1) ? out[(unsigned)(-pq[kk][j] - 2)] : -1;
    }
#pragma unroll
    for (int kk = 0; kk < G; kk++) {
      if (k0 + kk >= EPT) continue;
      const int k = k0 + kk, i = tid + k * AT_NT;
      unsigned ref[4], c = c0[kk];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const bool late = i < n && pq[kk][j] < -1 && pv[kk][j] >= 0;
        const unsigned t = late ? (unsigned)size[(unsigned)pv[kk][j]] : 0u;
        c = t > c ? t : c;
        ref[j] = (i < n && pq[kk][j] < -1 && pv[kk][j] < 0) ? (unsigned)(-pv[kk][j] - 2) : AT_NONE;
      }
      rdiag[k] = ref[0] | (ref[2] << 16); rwn[k] = ref[3] | (ref[1] << 16);
      if (i < n) W[i] = at_pack(c, ref[3], ref[1]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  const unsigned long long t1 = wall_clock64();
  int sweeps = 0, steps = 0;
  bool settled = false;
  for (; sweeps < AT_SWEEPS && !settled; sweeps++) {
    for (int step = 0; step < nsteps; step++) {          // (a run lies in one row / one column and the links double their reach with every step: nsteps = log2 of the longer side, rounded up)
#pragma unroll
      for (int k = 0; k < EPT; k++) {
        const int i = tid + k * AT_NT;
        if (i >= n) continue;
        const ab_word w = W[i];
        unsigned xv = at_x(w), lw = at_lw(w), ln = at_ln(w);
        const unsigned x0 = xv, had = lw & ln;
        if (lw != AT_NONE) { const ab_word a = W[lw]; xv = at_x(a) > xv ? at_x(a) : xv; lw = at_lw(a); }
        if (ln != AT_NONE) { const ab_word a = W[ln]; xv = at_x(a) > xv ? at_x(a) : xv; ln = at_ln(a); }
        xv = at_relax(W, rdiag[k] & 0xffffu, xv);
        xv = at_relax(W, rdiag[k] >> 16, xv);
        if (xv != x0 || had != AT_NONE) W[i] = at_pack(xv, lw, ln);      // (nothing to store for a pixel without links whose value did not move)
      }
      steps++;
      __syncthreads();
    }
    // is x a fixed point of T = max(constant part, T of the four predecessors)?  (The constant part went in at the start and x only grows.)
    // If not: another sweep, links restored.
    bool bad = false;
    unsigned f[EPT];
#pragma unroll
    for (int k = 0; k < EPT; k++) {
      const int i = tid + k * AT_NT;
      f[k] = 0;
      if (i >= n) continue;
      const unsigned x0 = at_x(W[i]);
      unsigned m = x0;
      m = at_relax(W, rdiag[k] & 0xffffu, m);
      m = at_relax(W, rdiag[k] >> 16, m);
      m = at_relax(W, rwn[k] & 0xffffu, m);
      m = at_relax(W, rwn[k] >> 16, m);
      f[k] = m;
      bad = bad || m != x0;
    }
    settled = !__syncthreads_or(bad);
    if (!settled) {
#pragma unroll
      for (int k = 0; k < EPT; k++) { const int i = tid + k * AT_NT; if (i < n) W[i] = at_pack(f[k], rwn[k] & 0xffffu, rwn[k] >> 16); }
      __syncthreads();
    }
  }
  const unsigned long long t2 = wall_clock64();
  if (tid == 0) { status[2] = sweeps; status[3] = steps; if (!settled) status[0] = 1; }
  if (!settled) return;
  // labels: the first predecessor (NW, N, NE, W) whose T equals the pixel's own hands its label on - a pointer if that pixel is
  // undecided itself - else the best old label is it
  int L[EPT], p[EPT];
#pragma unroll
  for (int k = 0; k < EPT; k++) {
    const int i = tid + k * AT_NT;
    L[k] = 0; p[k] = 0;
    if (i >= n) continue;
    const int4 *rec = (const int4 *)(list + (size_t)i * AB_REC);
    const int4 a = rec[0], b = rec[1], d = rec[2];
    p[k] = a.x;
    int pl[4] = { a.w, b.x, b.y, b.z };
    unsigned ps[4] = { (unsigned)b.w, (unsigned)d.x, (unsigned)d.y, (unsigned)d.z };
    const unsigned ref[4] = { rdiag[k] & 0xffffu, rwn[k] >> 16, rdiag[k] >> 16, rwn[k] & 0xffffu };      // NW, N, NE, W
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (pl[j] < -1 && ref[j] == AT_NONE) { pl[j] = out[(unsigned)(-pl[j] - 2)]; ps[j] = (unsigned)size[(unsigned)pl[j]]; }     // (decided by its own tile, see above)
    const unsigned T = at_x(W[i]);
    int res = a.y;
    bool found = false;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (found) continue;
      if (ref[j] != AT_NONE) { if (at_x(W[ref[j]]) == T) { res = -(int)ref[j] - 1; found = true; } }
      else if (ps[j] == T) { res = pl[j]; found = true; }       // (T >= 1: a pixel outside the frame, size 0, never matches)
    }
    L[k] = res;
  }
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  const unsigned long long t3 = wall_clock64();
  int *Lp = (int *)W;
#pragma unroll
  for (int k = 0; k < EPT; k++) { const int i = tid + k * AT_NT; if (i < n) Lp[i] = L[k]; }
  __syncthreads();
  for (int step = 0; step < 16; step++) {           // (pointer chains may wind through rows and columns: until none is left)
    bool open = false;
#pragma unroll
    for (int k = 0; k < EPT; k++) {
      const int i = tid + k * AT_NT;
      if (i >= n || L[k] >= 0) continue;
      L[k] = Lp[-L[k] - 1];            // the predecessor's label, or the pointer it still follows
      Lp[i] = L[k];
      open = open || L[k] < 0;
    }
    if (!__syncthreads_or(open)) break;
  }
#pragma unroll
  for (int k = 0; k < EPT; k++) { const int i = tid + k * AT_NT; if (i < n) out[(unsigned)p[k]] = L[k]; }
  if (tid == 0) { status[4] = (int)(t1 - t0); status[5] = (int)(t2 - t1); status[6] = (int)(t3 - t2); status[7] = (int)(wall_clock64() - t3); }
}

__global__ __launch_bounds__(AT_NT) void k_absorb_tail(int *__restrict__ out, const int *__restrict__ list, int reccap, const int *__restrict__ count, const int *__restrict__ size, int nsteps, int *status, size_t zs) {
  RD_ZSHIFT(zs, out, list, count, size, status);
  extern __shared__ __attribute__((aligned(16))) ab_word at_lds[];
  const int n = *count;
  if (threadIdx.x == 0) { status[1] = n; status[0] = n > reccap ? 1 : 0; status[2] = 0; }
  if (n == 0 || n > reccap) return;
  if (n <= 2 * AT_NT) at_body<2>(at_lds, out, list, n, size, nsteps, status);
  else if (n <= 3 * AT_NT) at_body<3>(at_lds, out, list, n, size, nsteps, status);
  else if (n <= 4 * AT_NT) at_body<4>(at_lds, out, list, n, size, nsteps, status);
  else if (n <= 5 * AT_NT) at_body<5>(at_lds, out, list, n, size, nsteps, status);
  else if (n <= 6 * AT_NT) at_body<6>(at_lds, out, list, n, size, nsteps, status);
  else if (n <= 8 * AT_NT) at_body<8>(at_lds, out, list, n, size, nsteps, status);
  else at_body<16>(at_lds, out, list, n, size, nsteps, status);
}

// (rc:373-390, the region-boundary marks, are computed inside the labelling kernel: rd_k_label.hip, k_label_tile<true>)

// ------------------------------------------------------------------------------------------------ voting
// rc:426-464 in its canonical (raster) reading, SURVEY.md H9: a table slot is owned by the first pixel in raster order
// that touches it; every pixel whose segment id equals the owner's widens the slot's box by its own position, except
// that the owning pixel's very first touch does not.  Two launches: claim (atomicMin of pixel index per slot), then
// box (atomicMax), both over the pixels that carry a segment id.
__device__ __forceinline__ unsigned ls_slot(int id, int bid, int nentry) { return (((unsigned)id * (unsigned)bid) & 0x7fffffffu) % (unsigned)nentry; }

// sparse: one thread per chain pixel left by the polyline stage (raster-ordered compact list, so the compact index orders
// pixels like the pixel index does)
// The RB_MAX smallest distinct boundary ids greater than `floor` inside a 7x7 window, ascending, in registers (unused
// entries 0x7fffffff); returns whether larger ids remain.  Callers loop: windows with more ids than fit (rare) simply take
// another pass with floor = the largest id of the previous one, so every pass issues its dependent loads together.
#define RB_MAX 8
#define RB_NONE 0x7fffffff
__device__ __forceinline__ bool rb_collect(const int (&win)[49], int floor, int (&bs)[RB_MAX]) {
  // repeated minimum extraction: pass q finds the smallest id above the one of pass q-1; windows hold one to three distinct ids
  // as a rule, and the passes stop as soon as no lane of the wave finds another one
#pragma unroll
  for (int q = 0; q < RB_MAX; q++) bs[q] = RB_NONE;
  int cur = floor;
#pragma unroll
  for (int q = 0; q < RB_MAX; q++) {
    int m = RB_NONE;
#pragma unroll
    for (int k = 0; k < 49; k++) { const int b = win[k]; m = (b > cur && b < m) ? b : m; }
    bs[q] = m;
    cur = m;                              // (RB_NONE once exhausted: nothing is larger)
    if (!__any(m != RB_NONE)) break;
  }
  bool more = false;
  if (__any(bs[RB_MAX - 1] != RB_NONE)) {
    const int last = bs[RB_MAX - 1];
#pragma unroll
    for (int k = 0; k < 49; k++) more = more || (last != RB_NONE && win[k] > last);
  }
  return more;
}

// how often each collected id occurs in the window (entries without an id: 0)
__device__ __forceinline__ void rb_counts(const int (&win)[49], const int (&bs)[RB_MAX], int (&cnt)[RB_MAX]) {
#pragma unroll
  for (int q = 0; q < RB_MAX; q++) cnt[q] = 0;
#pragma unroll
  for (int q = 0; q < RB_MAX; q++) {
    if (!__any(bs[q] != RB_NONE)) break;
    int c = 0;
#pragma unroll
    for (int k = 0; k < 49; k++) c += win[k] == bs[q] ? 1 : 0;
    cnt[q] = bs[q] != RB_NONE ? c : 0;
  }
}

// slots that receive their first claim are appended to `tlist` (tlist[0] = count) so that the next frame can undo exactly
// these instead of clearing the whole 20-bytes-per-entry table
__device__ __forceinline__ void tlist_append(int *tlist, bool first, unsigned slot) {
  const unsigned long long m = __ballot(first);
  if (m) {
    const int lane = __lane_id(), leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(&tlist[0], __popcll(m));
    base = __shfl(base, leader);
    if (first) tlist[1 + base + __popcll(m & ((1ull << lane) - 1))] = (int)slot;
  }
}

// (the three vote kernels and the sampling kernel work on frame blockIdx.z of a batch: see rdk::PolyFrame)
#define RD_VFRAME const rdk::PolyFrame &FRM = FRS.f[blockIdx.z]; const rdk::PolyScratch &s = FRM.ps; (void)s
// -DRD_BOUNDARY_FLATTEN=0 (tuning builds, rd_kernels.h): the boundary-component plane arrives as the union-find forest the border kernel left (a pixel holds its tile's root,
// tile roots point towards the component's smallest pixel), and its readers - the 7x7 windows of the chain pixels, the 15 probes per segment - walk to the roots themselves.
__device__ __forceinline__ int boundary_root(const int *__restrict__ boundary, int l) {
#if !RD_BOUNDARY_FLATTEN
  if (l > 0) { int n = boundary[l]; while (n != l) { l = n; n = boundary[l]; } }
#endif
  return l;
}

__global__ __launch_bounds__(1024) void k_reduce_clean(const rdk::PolyFrames FRS) {
  RD_VFRAME;
  int *table = FRM.table, *claim = FRM.claim, *tlist = FRM.tlist;
  const int n = tlist[0];
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const int slot = tlist[1 + j];
    int *e = table + (size_t)slot * 5;
    e[0] = 0; e[1] = 0; e[2] = 0; e[3] = 0; e[4] = 0;
    claim[slot] = 0x7f7f7f7f;
  }
  __syncthreads();
  if (threadIdx.x == 0) tlist[0] = 0;
}

// (claims are aggregated per block like the box updates below: slot -> smallest claiming pixel in an LDS hash, one atomicMin per
//  slot and block at the end; the one that finds the slot unclaimed appends it to tlist)
#define CA_T 1024
__global__ __launch_bounds__(256) void k_reduce_claim(const rdk::PolyFrames FRS, int iw, int ih, int nentry) {
  RD_VFRAME;
  int *claim = FRM.claim, *tlist = FRM.tlist;
  const int *__restrict__ boundary = FRM.boundary;
  __shared__ int keys[CA_T], vals[CA_T];
  const int nlive = s.ctr[24];
  if ((int)(blockIdx.x * blockDim.x) >= nlive) return;       // (whole block without pixels)
  for (int t = threadIdx.x; t < CA_T; t += 256) { keys[t] = -1; vals[t] = 0x7f7f7f7f; }
  __syncthreads();
  const int stride = gridDim.x * blockDim.x;
  for (int j0 = blockIdx.x * blockDim.x; j0 < nlive; j0 += stride) {    // whole waves iterate together (ballots inside)
    const int j = j0 + threadIdx.x;
    const int i = j < nlive ? s.live[j] : 0;
    const int id = j < nlive ? s.id[i] : 0;
    const bool act = id > 0;
    const int p0 = act ? s.pos[i] : 0, x = p0 % iw, y = p0 / iw;
    int win[49];
#pragma unroll
    for (int k = 0; k < 49; k++) {
      const int xx = x + k % 7 - 3, yy = y + k / 7 - 3;
      win[k] = (act && xx >= 0 && xx < iw && yy >= 0 && yy < ih) ? boundary[yy * iw + xx] : 0;
    }
#if !RD_BOUNDARY_FLATTEN
    {   // the cells' walks to their components' roots take their steps together (most cells hold -1 or a root already)
      int nx[49];
      bool moving = false;
#pragma unroll
      for (int k = 0; k < 49; k++) { nx[k] = win[k] > 0 ? boundary[win[k]] : win[k]; moving = moving || nx[k] != win[k]; }
      while (__any(moving)) {
        moving = false;
#pragma unroll
        for (int k = 0; k < 49; k++) if (nx[k] != win[k]) { win[k] = nx[k]; nx[k] = boundary[win[k]]; moving = moving || nx[k] != win[k]; }
      }
    }
#endif
    int floor = 0;
    bool more = true;
    while (__any(more)) {               // one pass for all but a few pixels
      int bs[RB_MAX];
      const bool again = more && rb_collect(win, floor, bs);
      unsigned slot[RB_MAX];
#pragma unroll
      for (int q = 0; q < RB_MAX; q++) slot[q] = ls_slot(id, bs[q] == RB_NONE ? 0 : bs[q], nentry);
      if (floor == 0 && j < nlive) {
        // k_reduce_box visits the same slots: they are left for it (RB_MAX ints per chain pixel in the neighbour table of the
        // polyline stage, which is dead by now); -1: no slot, first entry -2: the window holds more ids than fit - start over
        // an entry also carries, in bits 28-29, how many cells of the window hold its id (capped at 3): with that the box kernel
        // can tell "the claiming pixel touched the slot at least twice" without reading the window again
        int cnt[RB_MAX], e[RB_MAX];
        rb_counts(win, bs, cnt);
#pragma unroll
        for (int q = 0; q < RB_MAX; q++) e[q] = bs[q] == RB_NONE ? -1 : (int)(slot[q] | ((unsigned)(cnt[q] < 3 ? cnt[q] : 3) << 28));
        if (again) e[0] = -2;
        int4 *rec = (int4 *)(s.nbr + (size_t)j * RB_MAX);
        rec[0] = make_int4(e[0], e[1], e[2], e[3]); rec[1] = make_int4(e[4], e[5], e[6], e[7]);
        s.lab[j] = id; s.alive[j] = p0;       // (dead arrays of the polyline stage as well: saves the box kernel a level of gathers)
      }
#pragma unroll
      for (int q = 0; q < RB_MAX; q++) {
        if (!(more && bs[q] != RB_NONE)) continue;
        unsigned h = (slot[q] * 2654435761u) >> 22;
        int probes = 0;
        for (;;) {
          const int kprev = atomicCAS(&keys[h], -1, (int)slot[q]);
          if (kprev == -1 || kprev == (int)slot[q]) { atomicMin(&vals[h], i); break; }
          h = (h + 1) & (CA_T - 1);
          if (++probes == 32) {         // table full of other slots: straight to memory
            if (atomicMin(&claim[slot[q]], i) == 0x7f7f7f7f) tlist[1 + atomicAdd(&tlist[0], 1)] = (int)slot[q];
            break;
          }
        }
      }
      floor = bs[RB_MAX - 1];
      more = again;
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < CA_T; t += 256) {       // (CA_T is a multiple of 256: whole waves call tlist_append together)
    const int key = keys[t];
    bool first = false;
    if (key != -1) { const int v = vals[t]; if (v < ld_agent(&claim[key])) first = atomicMin(&claim[key], v) == 0x7f7f7f7f; }
    tlist_append(tlist, first, (unsigned)key);
  }
}

// Block-local aggregation of the box updates: hundreds of pixels of one segment widen the same slot, and atomics on one address
// are served one after the other (measured: the longest segment of a frame set the duration of the kernel).  The block keeps a
// small hash (slot -> the four maxima) in LDS and touches each of its slots once in global memory at the end.
#define BA_T 512
struct BoxAgg { int keys[BA_T]; int vals[BA_T * 4]; };
__device__ __forceinline__ void boxagg_add(BoxAgg &A, int *table, unsigned slot, int v1, int v2, int v3, int v4) {
  unsigned h = (slot * 2654435761u) >> 23;
  for (int probes = 0; probes < 16; probes++) {
    const int kprev = atomicCAS(&A.keys[h], -1, (int)slot);
    if (kprev == -1 || kprev == (int)slot) {
      atomicMax(&A.vals[h * 4 + 0], v1); atomicMax(&A.vals[h * 4 + 1], v2); atomicMax(&A.vals[h * 4 + 2], v3); atomicMax(&A.vals[h * 4 + 3], v4);
      return;
    }
    h = (h + 1) & (BA_T - 1);
  }
  int *e = table + (size_t)slot * 5;      // table full of other slots: straight to memory
  atomicMax(&e[1], v1); atomicMax(&e[2], v2); atomicMax(&e[3], v3); atomicMax(&e[4], v4);
}

// widening for up to RB_MAX slots of one chain pixel (valid[q]: slot q is in use); `touches(slot)` counts how often the pixel's
// window maps to a slot and is only evaluated for a pixel that holds the claim itself
template <typename Touches>
__device__ __forceinline__ void box_vote(BoxAgg &A, int *table, const int *__restrict__ claim, const int *__restrict__ ids, const unsigned (&slot)[RB_MAX], const bool (&valid)[RB_MAX],
                                         int i, int id, int x, int y, int iw, int ih, Touches touches) {
  // Per distinct slot: the claiming pixel's first touch only claims (rc:449-456), every other touch widens the box; max
  // is idempotent, so "widen once if the slot's owner carries our id and we touched it often enough" is the same.
  int owner[RB_MAX], oid[RB_MAX];
#pragma unroll
  for (int q = 0; q < RB_MAX; q++) owner[q] = claim[slot[q]];          // (unused slots map to entry 0: a valid address)
#pragma unroll
  for (int q = 0; q < RB_MAX; q++) oid[q] = valid[q] ? ids[owner[q]] : -1;
#pragma unroll
  for (int q = 0; q < RB_MAX; q++) {
    if (!valid[q] || oid[q] != id) continue;
    if (owner[q] == i) {
      table[(size_t)slot[q] * 5] = id;            // the slot's id is its claimant's (exactly one pixel holds the claim)
      if (touches(slot[q]) < 2) continue;         // our first touch does not count
    }
    boxagg_add(A, table, slot[q], iw - x, x, ih - y, y);
  }
}

__global__ __launch_bounds__(256) void k_reduce_box(const rdk::PolyFrames FRS, int iw, int ih, int nentry) {
  RD_VFRAME;
  int *table = FRM.table;
  const int *__restrict__ claim = FRM.claim;
  const int *__restrict__ boundary = FRM.boundary;
  __shared__ BoxAgg A;
  const int nlive = s.ctr[24];
  if ((int)(blockIdx.x * blockDim.x) >= nlive) return;       // (whole block without pixels)
  for (int t = threadIdx.x; t < BA_T; t += 256) { A.keys[t] = -1; A.vals[t * 4] = 0; A.vals[t * 4 + 1] = 0; A.vals[t * 4 + 2] = 0; A.vals[t * 4 + 3] = 0; }
  __syncthreads();
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < nlive; j += gridDim.x * blockDim.x) {
    const int i = s.live[j];
    const int4 *rec = (const int4 *)(s.nbr + (size_t)j * RB_MAX);      // left by k_reduce_claim, like the pixel's id and position
    const int4 r0 = rec[0], r1 = rec[1];
    const int id = s.lab[j];
    const int p0 = s.alive[j], x = p0 % iw, y = p0 / iw;
    if (id <= 0) continue;
    if (r0.x != -2) {
      const int rs[RB_MAX] = { r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w };
      unsigned slot[RB_MAX];
      bool valid[RB_MAX];
      int cnt[RB_MAX];
#pragma unroll
      for (int q = 0; q < RB_MAX; q++) { valid[q] = rs[q] >= 0; slot[q] = valid[q] ? (unsigned)rs[q] & 0x0fffffffu : 0u; cnt[q] = valid[q] ? rs[q] >> 28 : 0; }
      box_vote(A, table, claim, s.id, slot, valid, i, id, x, y, iw, ih, [&](unsigned sl) {
        int touches = 0;          // cells of the window whose id maps to this slot (several ids may: same slot value in the record)
#pragma unroll
        for (int q = 0; q < RB_MAX; q++) touches += (valid[q] && slot[q] == sl) ? cnt[q] : 0;
        return touches;
      });
      continue;
    }
    // more distinct boundary ids in the window than one pass holds (rare): collect them again, pass by pass
    int win[49];
#pragma unroll
    for (int k = 0; k < 49; k++) {
      const int xx = x + k % 7 - 3, yy = y + k / 7 - 3;
      win[k] = boundary_root(boundary, (xx >= 0 && xx < iw && yy >= 0 && yy < ih) ? boundary[yy * iw + xx] : 0);
    }
    int floor = 0;
    bool more = true;
    while (more) {
      int bs[RB_MAX];
      more = rb_collect(win, floor, bs);
      unsigned slot[RB_MAX];
      bool valid[RB_MAX];
#pragma unroll
      for (int q = 0; q < RB_MAX; q++) { valid[q] = bs[q] != RB_NONE; slot[q] = ls_slot(id, valid[q] ? bs[q] : 0, nentry); }
      box_vote(A, table, claim, s.id, slot, valid, i, id, x, y, iw, ih, [&](unsigned sl) {
        int touches = 0;
        for (int k = 0; k < 49; k++) touches += (win[k] > 0 && ls_slot(id, win[k], nentry) == sl);
        return touches;
      });
      floor = bs[RB_MAX - 1];
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < BA_T; t += 256) {
    const int key = A.keys[t];
    if (key == -1) continue;
    int *e = table + (size_t)key * 5;
#pragma unroll
    for (int k = 0; k < 4; k++) { const int v = A.vals[t * 4 + k]; if (v > ld_agent(&e[1 + k])) atomicMax(&e[1 + k], v); }
  }
}

// ------------------------------------------------------------------------------------------------ result sampling
// rh:1066-1098: 3 points along each valid segment x 5 offsets along its normal; the boundary id under each probe and
// the voting-table slot of (segment, boundary id) are gathered on the device so that only n*15*6 ints travel to the
// host instead of the 4N-byte plane and the 16N-byte table.  Double precision, same operation order as the host code.
struct ls_rec { float x0, y0, x1, y1; int startIndex, endIndex, leftPtr, rightPtr, startCount, endCount, maxDist, polyid, npix, level; };

// The kernel also assembles the block that travels to the host in ONE copy (pack): [0,32) polyline counters, [32,52) region
// round flags, [52,60) absorption status, [64, 64 + 14 * pack_records) segment records 0.., then 90 ints of probes per record.
__global__ void k_sample_segments(const rdk::PolyFrames FRS, int max_records, int iw, int ih, int nentry, int pack_records) {
  RD_VFRAME;
  int *__restrict__ out = FRM.probes;
  const ls_rec *__restrict__ ls = (const ls_rec *)FRM.lslist;
  const int *__restrict__ boundary = FRM.boundary;
  const int *__restrict__ table = FRM.table;
  int *__restrict__ pack = FRM.pack;
  const int *__restrict__ polyctr = s.ctr;
  const int *__restrict__ rflags = FRM.rflags;
  const int n = *(const int *)ls;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (pack) {
    if (t < 32) pack[t] = polyctr[t];
    else if (t < 52) pack[t] = rflags[t - 32];        // (the flags of the first 20 launches of the region merge)
    else if (t < 60) pack[t] = rflags[RD_REGION_STATUS_AT + t - 52];   // (the status words of the absorption)
    if (t < 14) pack[64 + t] = ((const int *)ls)[t];        // header record
  }
  const int i = t / 15 + 1, k = t % 15;
  if (i > n || i >= max_records) return;
  int *o = out + (size_t)(i * 15 + k) * 6;
  const bool packed = pack && i < pack_records;
  if (packed && k < 14) pack[64 + i * 14 + k] = ((const int *)ls)[i * 14 + k];   // the record itself (14 ints), one int per probe thread
  int segid = 0;
  if (ls[i].polyid != 0) {
    int sx, sy;
    if (rdp_probe_pixel(ls[i].x0, ls[i].y0, ls[i].x1, ls[i].y1, k, iw, ih, &sx, &sy)) segid = boundary_root(boundary, boundary[sx + sy * iw]);
  }
  int v[6] = { segid, 0, 0, 0, 0, 0 };
  if (segid > 0) {
    const unsigned slot = ls_slot(i, segid, nentry);
    const int *e = table + (size_t)slot * 5;
    v[1] = e[0]; v[2] = e[1]; v[3] = e[2]; v[4] = e[3]; v[5] = e[4];
  }
#pragma unroll
  for (int q = 0; q < 6; q++) o[q] = v[q];
  if (packed) {
    int *po = pack + 64 + pack_records * 14 + (size_t)(i * 15 + k) * 6;
#pragma unroll
    for (int q = 0; q < 6; q++) po[q] = v[q];
  }
}

}  // namespace

namespace rdk {

// bits: ih * ceil(iw / 64) * 2 words ("counted", "curve end" per 64 pixels), from the strong mask's bit plane
void junction_bits(hipStream_t s, unsigned long long *bits, const unsigned long long *strong, int iw, int ih, int nz, size_t zs) {
  const int wpr = cdiv(iw, 64);
  hipLaunchKernelGGL(k_junction_bits, dim3(cdiv(wpr * ih, 256), 1, nz), dim3(256), 0, s, bits, strong, iw, ih, wpr, zs);
}
void blblur_extents(hipStream_t s, uint16_t *ext, const int8_t *edge, int iw, int ih, int nz, size_t zs) {
  hipLaunchKernelGGL(k_blblur_extents, dim3(cdiv(iw, 64), cdiv(ih, BE_ROWS), nz), dim3(64, 4), 0, s, ext, edge, iw, ih, zs);
}
void blblur_pair(hipStream_t s, uint32_t *out, const uint16_t *ext, const uint32_t *in, int iw, int ih, int nz, size_t zs) {
  hipLaunchKernelGGL(k_blblur_pair, dim3(rd_tile_blocks(cdiv(iw, 64), cdiv(ih, BQ_ROWS), nz)), dim3(64, 8), 0, s, out, ext, in, iw, ih, zs, rd_gdim(cdiv(iw, 64), cdiv(ih, BQ_ROWS), nz));
}
// fills the quantisation tables of the current device (once per device, before its first frame; the caller synchronises)
void quant_lut_init(hipStream_t s) { hipLaunchKernelGGL(k_quant24_lut, dim3(20), dim3(256), 0, s); }
void despeckle(hipStream_t s, uint32_t *out, const uint32_t *in, const float *edge, int iw, int ih, int quantize24, int nz, size_t zs) {
  if (quantize24) hipLaunchKernelGGL(k_despeckle<24>, dim3(rd_tile_blocks(cdiv(iw, 64), cdiv(ih, DS_ROWS), nz)), dim3(64, 4), 0, s, out, in, edge, iw, ih, zs, rd_gdim(cdiv(iw, 64), cdiv(ih, DS_ROWS), nz));
  else hipLaunchKernelGGL(k_despeckle<0>, dim3(rd_tile_blocks(cdiv(iw, 64), cdiv(ih, DS_ROWS), nz)), dim3(64, 4), 0, s, out, in, edge, iw, ih, zs, rd_gdim(cdiv(iw, 64), cdiv(ih, DS_ROWS), nz));
}
// out: the merge mask as a bit plane (ih * ceil(iw / 64) words); bits: what junction_bits() left
void merge_mask(hipStream_t s, unsigned long long *out, const unsigned long long *bits, int iw, int ih, int nz, size_t zs) {
  const int wpr = cdiv(iw, 64);
  hipLaunchKernelGGL(k_mm_gather, dim3(wpr, cdiv(ih, MM_ROWS), nz), dim3(64, 4), 0, s, out, bits, iw, ih, wpr, zs);
}

// scratch: 3*N + 256 ints ([N, N + 96): a flag per launch and the absorption's status words, then the allowed-direction bytes; [2N, 3N): the second label plane of the rounds).
// ROUNDS: the number of launches, even (the last one writes `label`); launches after one that changed nothing return at once.
// *marked <- 1: `label` holds the rounds' words (label << 3 | mark), which region_size turns into plain labels.
void region_merge(hipStream_t s, int *label, int *scratch, const int *pix, const unsigned long long *mask, const unsigned long long *edge, int iw, int ih, int ROUNDS, int *size_out, int *marked, int nz, size_t zs) {
  const int n = iw * ih;
  if (ROUNDS < 2 || (ROUNDS & 1) || ROUNDS > RD_REGION_MAX_LAUNCHES) { fprintf(stderr, "region_merge: the number of launches must be even, 2..%d (got %d)\n", RD_REGION_MAX_LAUNCHES, ROUNDS); abort(); }
  int *flags = scratch + n;
  u64 *allow = (u64 *)(((uintptr_t)(flags + RR_NFLAGS) + 7) & ~(uintptr_t)7);      // ceil(ih / 8) * iw words: the allow bytes of 8 rows per column (k_region_init)
  int *A = label, *B = scratch + 2 * (size_t)n;   // the two planes of the rounds; the result is in A
  hipLaunchKernelGGL(k_region_init, dim3(rd_tile_blocks(cdiv(iw, 64), cdiv(ih, RI_ROWS), nz)), dim3(64, 4), 0, s, A, B, allow, pix, mask, edge, iw, ih, cdiv(iw, 64), flags, size_out, zs, rd_gdim(cdiv(iw, 64), cdiv(ih, RI_ROWS), nz));
  const dim3 grid(cdiv(iw, 64), cdiv(ih, RR_TY * RR_PX), nz);
  for (int r = 1; r < ROUNDS; r++) {       // (launch 0 was evaluated by k_region_init)
    const dim3 lg(rd_tile_blocks((int)grid.x, (int)grid.y, (int)grid.z));
    const int gdim = rd_gdim((int)grid.x, (int)grid.y, (int)grid.z);
    int *X = (r & 1) ? B : A, *Y = (r & 1) ? A : B;
    if (r == 1) hipLaunchKernelGGL(k_region_round<1>, lg, dim3(64, RR_TY), 0, s, X, Y, (const u64 *)allow, iw, ih, flags, r, zs, gdim);
    else if (r < RR_DEEP) hipLaunchKernelGGL(k_region_round<2>, lg, dim3(64, RR_TY), 0, s, X, Y, (const u64 *)allow, iw, ih, flags, r, zs, gdim);
    else hipLaunchKernelGGL(k_region_round<3>, lg, dim3(64, RR_TY), 0, s, X, Y, (const u64 *)allow, iw, ih, flags, r, zs, gdim);
  }
  if (marked) *marked = 1;
}

void region_size(hipStream_t s, int *out, int *label, int n, int *zero_me, int marked, int nz, size_t zs) {
  hipLaunchKernelGGL(k_region_size, dim3(cdiv(n, 256 * RS_PER_THREAD), 1, nz), dim3(256), 0, s, out, label, n, zero_me, marked, zs);
}

// rc:348-371 as the reference's serial raster order evaluates it, exactly.  scratch: RD_D2_SCRATCH_INTS(N) ints, scratch[N] (the list
// counter) zeroed by the caller when count_is_zero; out must not alias in.  status (device, 3 ints): [0] != 0 <=> the two launches
// could not finish the frame (more than AT_CAP undecided pixels, or the tail did not settle): `out` then still holds undecided
// words (negative) and the caller runs despeckle2_slow(); [1] undecided pixels left by the tile kernel, [2] sweeps of the tail.
void despeckle2(hipStream_t s, int *out, const int *in, int *scratch, const int *size, int thre, int iw, int ih, int count_is_zero, int *status, int nz, size_t zs) {
  const int n = iw * ih;
  int *count = scratch + (size_t)n, *list = count + 16 + (size_t)n;       // (the list: the 3N ints of the slow path's work lists)
  const int reccap = (int)(3 * (size_t)n / AB_REC) < AT_CAP ? (int)(3 * (size_t)n / AB_REC) : AT_CAP;
  int nsteps = 1;
  while ((1 << nsteps) < (iw > ih ? iw : ih)) nsteps++;
  if (!count_is_zero) { if (nz != 1) { fprintf(stderr, "despeckle2: a group launch needs count_is_zero\n"); abort(); } (void)hipMemsetAsync(count, 0, sizeof(int), s); }
  hipLaunchKernelGGL(k_absorb_tile, dim3(rd_tile_blocks(cdiv(iw, AB_TW), cdiv(ih, AB_TH), nz)), dim3(AB_NT), 0, s, out, list, reccap, count, in, size, thre, iw, ih, zs, rd_gdim(cdiv(iw, AB_TW), cdiv(ih, AB_TH), nz));
  static std::atomic<unsigned> lds_set{0};
  set_max_lds_once((const void *)k_absorb_tail, (int)((AT_CAP + 1) * sizeof(ab_word)), lds_set);
  hipLaunchKernelGGL(k_absorb_tail, dim3(1, 1, nz), dim3(AT_NT), (AT_CAP + 1) * sizeof(ab_word), s, out, (const int *)list, reccap, (const int *)count, size, nsteps, status, zs);
}

// The same result by plain Jacobi rounds of the recurrence over work lists in global memory, two rounds per launch, until a launch
// changes nothing (the fixed point of the rounds is the serial result; chains of dependent pixels can be thousands long on frames
// that consist of small regions only, so this takes as many launches as it takes).  Synchronises `s` every few launches: for the
// frames the fast path gives up on, outside captured graphs.  Same scratch as despeckle2().
void despeckle2_slow(hipStream_t s, int *out, const int *in, int *scratch, const int *size, int thre, int iw, int ih) {
  const int n = iw * ih;
  int *tmp = scratch, *count = scratch + (size_t)n, *stamp = count + 16, *lists = stamp + (size_t)n;
  (void)hipMemsetAsync(count, 0, 16 * sizeof(int), s);
  // first round: tmp <- result, out <- input (both planes then agree on every pixel that is not in the list)
  hipLaunchKernelGGL(k_despeckle2_first, dim3(cdiv(iw, 64), cdiv(ih, D2_ROWS)), dim3(64, 4), 0, s, tmp, out, stamp, lists, count, in, size, thre, iw, ih);
  const int *cur = tmp;
  int r = 0;
  for (;;) {
    for (int k = 0; k < 8; k++, r++) {
      int *nxt = (r & 1) ? tmp : out;
      const int li = r % 3, lo = (r + 1) % 3, lz = (r + 2) % 3;
      hipLaunchKernelGGL(k_despeckle2_active, dim3(256), dim3(256), 0, s, nxt, cur, (const int *)(lists + (size_t)li * n), (const int *)(count + li),
                         lists + (size_t)lo * n, count + lo, count + lz, count + 3, stamp, r + 1, in, size, thre, iw, ih);
      cur = nxt;
    }
    int pending = 0;      // the list the last launch produced
    (void)hipMemcpyAsync(&pending, count + (r % 3), sizeof(int), hipMemcpyDeviceToHost, s);
    (void)hipStreamSynchronize(s);
    if (pending == 0) break;
    // (a dependency chain moves at least one pixel right or one row down per link: iw + 2 * ih bounds its length, and every launch settles two links)
    if (r > iw + 2 * ih + 16) { fprintf(stderr, "despeckle2_slow: no fixed point after %d launches (internal error)\n", r); abort(); }
  }
  // (launches with an odd number write `tmp`, and the last one had: the result moves to `out`)
  if (cur != out) (void)hipMemcpyAsync(out, cur, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, s);
}

// table: nentry*5 ints, claim: nentry ints, tlist: nentry+1 ints; all three are set up once by reduce_ls_init and kept
// clean between frames by undoing exactly the touched slots
void reduce_ls_init(hipStream_t s, int *table, int *claim, int *tlist, int nentry) {
  (void)hipMemsetAsync(table, 0, sizeof(int) * 5 * (size_t)nentry, s);
  (void)hipMemsetAsync(claim, 0x7f, sizeof(int) * (size_t)nentry, s);
  (void)hipMemsetAsync(tlist, 0, sizeof(int), s);
}

void reduce_ls(hipStream_t s, const PolyFrame *frames_host, int nb, int iw, int ih, int nentry, int tables_are_clean) {
  const PolyFrames frames = pack_frames(frames_host, nb);
  if (!tables_are_clean) hipLaunchKernelGGL(k_reduce_clean, dim3(1, 1, nb), dim3(1024), 0, s, frames);       // undo the previous use
  hipLaunchKernelGGL(k_reduce_claim, dim3(512, 1, nb), dim3(256), 0, s, frames, iw, ih, nentry);
  hipLaunchKernelGGL(k_reduce_box, dim3(512, 1, nb), dim3(256), 0, s, frames, iw, ih, nentry);
}

void sample_segments(hipStream_t s, const PolyFrame *frames_host, int nb, int max_records, int iw, int ih, int nentry, int pack_records) {
  const PolyFrames frames = pack_frames(frames_host, nb);
  const int threads = max_records * 15;
  hipLaunchKernelGGL(k_sample_segments, dim3(cdiv(threads, 256), 1, nb), dim3(256), 0, s, frames, max_records, iw, ih, nentry, pack_records);
}

}  // namespace rdk
