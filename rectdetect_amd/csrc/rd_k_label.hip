// rectdetect-mi355x: connected-component labelling and per-label strength reduction for gfx950.
//
// The reference labels components with a fixed number of in-place min-propagation passes
// (oclimgutil.cl:495-538, oclimgutil.c:227-246: 1 + 10 launches, converged only by luck - SURVEY.md H4).
// Here the labelling is a run-based union-find that always converges: label = smallest pixel index of the
// 8-connected component of equal pixel value, -1 for pixels equal to the background value.
#include <stdlib.h>
#include "rd_device.h"
#include "rd_kernels.h"
#include "rd_tidy_tile.h"

namespace {

using namespace rd;

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
const dim3 block2(64, 4);
inline dim3 grid2(int iw, int ih) { return dim3(cdiv(iw, 64), cdiv(ih, 4)); }

// Phase 1 (one block per 64x32 tile, everything in LDS): runs inside each 64-pixel row segment (ballot of run starts +
// count-leading-zeros), unions between the rows of the tile, path compression; the tile's pixels leave pointing at the
// GLOBAL index of their tile-local root, which is the smallest index of the component's part inside the tile.
#define LT_W 64
#ifndef LT_H
#define LT_H 32
#endif
// (the walks re-read words other lanes lower meanwhile: relaxed atomic loads, workgroup scope - LDS reads as ds_read.  A `volatile` pointer instead loses its
//  address space: every step of every walk was a FLAT load with system-scope bits, several times the latency of the LDS instruction)
__device__ __forceinline__ int lt_ld(const int *lab, int a) { return __hip_atomic_load(lab + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int lt_find(const int *lab, int a) {
  int l = lt_ld(lab, a);
  while (l != a) { a = l; l = lt_ld(lab, a); }
  return a;
}
__device__ __forceinline__ void lt_union(int *lab, int a, int b) {
  for (;;) {
    a = lt_find(lab, a);
    b = lt_find(lab, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }
    const int old = atomicMin(&lab[a], b);
    if (old == a) return;
    a = old;
  }
}

#ifndef LT_TY
#define LT_TY 4
#endif
#ifndef LT_TY1
#define LT_TY1 LT_TY      // thread rows of the boundary variant (the tidy variant's tile code is written for four waves)
#endif
// LT_TY: thread rows per block (16 is ~20% faster run alone, but costs 50% more wave-cycles: worse with frames in flight)
#define LT_MP 68      // row pitch of the staged region tile of the boundary variant (64 + 2 x 2 cells of halo)
// BOUNDARY = false: the pixel values are read from `pix`.  BOUNDARY = true: they are the region-boundary marks of oclrect.cl:373-390,
// computed here from the region plane `src` (a pixel of the interior whose 5x5 window holds another label carries its own label,
// every other pixel -1; evaluated separably like this: hu = "the five cells x-2..x+2 of a row equal the one at x", window uniform
// iff the five cells of the centre column equal the centre and their rows are uniform) and also written to `pix_out` for the
// border kernel - one launch and one pass over the plane less than marking first and labelling then.
// SRC == 2: the pixel values are the rect-variant edge tidy of the NMS response `nms` (rd_tidy_tile.h), computed here and written to
// mask0 / pix_out (and zero_plane cleared) as k_rect_tidy would.
template <int SRC, int TY>
__global__ __launch_bounds__(64 * TY) void k_label_tile(int *__restrict__ label, const int *__restrict__ pix, int bgc, int iw, int ih, int *__restrict__ pix_out,
                                                    const float *__restrict__ nms, int *__restrict__ mask0, int *__restrict__ zero_plane, size_t zs, int gdim) {
  constexpr bool BOUNDARY = SRC == 1;
  const rd_tile rd_b = rd_block_tile(gdim);
  if (rd_b.x < 0) return;
  RD_ZSHIFTZ(rd_b.z, zs, label, pix, pix_out, nms, mask0, zero_plane);
  __shared__ int lab[LT_W * LT_H];
  __shared__ int pv[LT_W * LT_H];
  const int tx = threadIdx.x, x = rd_b.x * LT_W + tx, y0 = rd_b.y * LT_H;
  const bool xin = x < iw;
  int v00;
  bool uniform = true;
  int pv8[LT_H / TY];            // this thread's pixels, requested together (one wait for memory instead of one per row)
  if constexpr (SRC == 2) {
    __shared__ __align__(16) uint8_t A[(LT_H + 2 * TD_M) * TD_P], B[(LT_H + 2 * TD_M) * TD_P];
    rect_tidy_tile<LT_H>(A, B, rd_b.x * LT_W, y0, threadIdx.y * 64 + tx, nms, mask0, pix_out, zero_plane, iw, ih, pv8);
    __shared__ int s_t00;            // the value of the tile's first pixel, for the uniform-tile test below
    if (threadIdx.y == 0 && tx == 0) s_t00 = pv8[0];
    __syncthreads();
    v00 = s_t00;
  } else if constexpr (!BOUNDARY) {
    v00 = pix[(size_t)y0 * iw + rd_b.x * LT_W];   // the tile's first pixel is always inside the frame
#pragma unroll
    for (int k = 0; k < LT_H / TY; k++) {
      const int y = y0 + threadIdx.y + k * TY;
      pv8[k] = pix[(xin && y < ih) ? y * iw + x : 0];
    }
  } else {
    __shared__ int t[(LT_H + 4) * LT_MP];
    __shared__ uint8_t hu[(LT_H + 4) * 64];
    const int x0 = rd_b.x * LT_W, tid = threadIdx.y * 64 + tx;
    const int r00 = pix[(size_t)y0 * iw + x0];
    bool flat = true;
    {
      // rows of the staged tile by waves, columns 0..63 by lanes, the 4 columns left over a cell of the first 4 (LT_H + 4) threads: no division per cell,
      // all loads of a thread in flight together
      constexpr int NR = (LT_H + 4 + TY - 1) / TY;
      static_assert(LT_MP == 68 && 4 * (LT_H + 4) <= 64 * TY, "four columns beyond the lanes, one cell per thread");
      int v[NR + 1];
      bool ok[NR + 1];
      const int xl = x0 - 2 + tx;
      const bool cok = xl >= 0 && xl < iw;
#pragma unroll
      for (int i = 0; i < NR; i++) {
        const int r = threadIdx.y + TY * i, yy = y0 - 2 + r;
        ok[i] = r < LT_H + 4 && cok && yy >= 0 && yy < ih;
        v[i] = pix[ok[i] ? yy * iw + xl : 0];
      }
      const int sr = tid >> 2, sc = 64 + (tid & 3);
      {
        const int xx = x0 - 2 + sc, yy = y0 - 2 + sr;
        ok[NR] = sr < LT_H + 4 && xx < iw && yy >= 0 && yy < ih;
        v[NR] = pix[ok[NR] ? yy * iw + xx : 0];
      }
#pragma unroll
      for (int i = 0; i < NR; i++) {
        const int r = threadIdx.y + TY * i;
        if (r < LT_H + 4) { flat = flat && (!ok[i] || v[i] == r00); t[r * LT_MP + tx] = ok[i] ? v[i] : 0; }
      }
      if (sr < LT_H + 4) { flat = flat && (!ok[NR] || v[NR] == r00); t[sr * LT_MP + sc] = ok[NR] ? v[NR] : 0; }
    }
    if (__syncthreads_and(flat)) {
      // no differing cell anywhere in reach: nothing is a boundary pixel, nothing to label
#pragma unroll
      for (int k = 0; k < LT_H / TY; k++) {
        const int y = y0 + threadIdx.y + k * TY;
        if (xin && y < ih) { pix_out[y * iw + x] = -1; label[y * iw + x] = -1; }
      }
      return;
    }
    for (int r = threadIdx.y; r < LT_H + 4; r += TY) {
      const int *row = t + r * LT_MP + tx + 2;
      const int c = row[0];
      hu[r * 64 + tx] = (row[-2] == c && row[-1] == c && row[1] == c && row[2] == c) ? 1 : 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < LT_H / TY; k++) {
      const int r = threadIdx.y + k * TY;
      const int y = y0 + r;
      int res = -1;
      if (xin && y < ih && x > 1 && y > 1 && x < iw - 2 && y < ih - 2) {
        const int c0 = t[(r + 2) * LT_MP + tx + 2];
        bool same = true;
#pragma unroll
        for (int dy = 0; dy < 5; dy++) same = same && t[(r + dy) * LT_MP + tx + 2] == c0 && hu[(r + dy) * 64 + tx] != 0;
        if (!same) res = c0;
      }
      pv8[k] = res;
      if (xin && y < ih) pix_out[y * iw + x] = res;
    }
    __shared__ int s_v00;            // the mark of the tile's first pixel, for the uniform-tile test below
    if (threadIdx.y == 0 && tx == 0) s_v00 = pv8[0];
    __syncthreads();
    v00 = s_v00;
  }
#pragma unroll
  for (int k = 0; k < LT_H / TY; k++) {
    const int r = threadIdx.y + k * TY;
    const int y = y0 + r;
    const bool valid = xin && y < ih;
    const int v = valid ? pv8[k] : 0;
    uniform = uniform && (!valid || v == v00);
    const int vl = __shfl_up(v, 1);
    const bool lvalid = __shfl_up((int)valid, 1) != 0;
    const bool same = valid && tx > 0 && lvalid && vl == v;
    const unsigned long long starts = __ballot(!same);
    const unsigned long long upto = starts & ((2ull << tx) - 1ull);
    const int start = 63 - __clzll((long long)upto);
    pv[r * LT_W + tx] = v;
    lab[r * LT_W + tx] = (!valid || v == bgc) ? -1 : r * LT_W + start;
  }
  // fast path: the whole tile holds one value (background, the inside of a large component): one component, no unions
  {
    if (__syncthreads_and(uniform)) {
      const int l = v00 == bgc ? -1 : y0 * iw + rd_b.x * LT_W;
#pragma unroll
      for (int r = threadIdx.y; r < LT_H; r += TY) {
        const int y = y0 + r;
        if (xin && y < ih) label[y * iw + x] = l;
      }
      return;
    }
  }
  // unions with the row above, inside the tile; a pixel only issues one when no pixel of its run is guaranteed to issue
  // an equivalent one (same case analysis as k_label_border below)
#pragma unroll
  for (int r = threadIdx.y; r < LT_H; r += TY) {
    if (r == 0) continue;
    const int q = r * LT_W + tx;
    if (lab[q] < 0) continue;
    const int v = pv[q];
    const bool wSame = tx > 0 && lab[q - 1] >= 0 && pv[q - 1] == v;
    const bool nSame = lab[q - LT_W] >= 0 && pv[q - LT_W] == v;
    const bool nwSame = tx > 0 && lab[q - LT_W - 1] >= 0 && pv[q - LT_W - 1] == v;
    if (nSame) {
      if (!(wSame && nwSame)) lt_union(lab, q, q - LT_W);
    } else {
      const bool neSame = tx < LT_W - 1 && lab[q - LT_W + 1] >= 0 && pv[q - LT_W + 1] == v;
      const bool eSame = tx < LT_W - 1 && lab[q + 1] >= 0 && pv[q + 1] == v;
      if (nwSame && !wSame) lt_union(lab, q, q - LT_W - 1);
      if (neSame && !eSame) lt_union(lab, q, q - LT_W + 1);
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = threadIdx.y; r < LT_H; r += TY) {
    const int y = y0 + r;
    if (!xin || y >= ih) continue;
    const int q = r * LT_W + tx;
    int l = lab[q];
    if (l >= 0) { l = lt_find(lab, l); l = (y0 + l / LT_W) * iw + rd_b.x * LT_W + l % LT_W; }
    label[y * iw + x] = l;
  }
}

// Phase 2: unions across tile borders, on the global array (roots: label[r] == r).  `horizontal` = 1: the pixels of rows
// y = 32k (k >= 1) against the row above; 0: the pixels of columns x = 64k (k >= 1) against column x-1 and the pixels of
// columns x = 64k-1 against their NE neighbour; lanes are consecutive pixels ALONG the border so that a lane whose
// (label, neighbour label) pair equals its predecessor's can leave the union to that lane.
// la, lb: label[p], label[q] (the tile roots: constant during this kernel unless roots themselves; any value when !want)
// seen (optional): a block-wide table of the (tile root, tile root) pairs somebody in the block has already taken on (LB_SEEN 64-bit slots in LDS, all ~0 at the start): along
// a border the background's runs alternate with the components that cross it, and every one of those runs asks for the SAME union of the two tiles' background roots - the
// walks to the roots are the kernel's time (chains of dependent loads), so a pair is walked once per block, by the first lane that claims it.  (Whoever claims a pair carries
// the union out before the kernel ends; unions commute and are idempotent, so it does not matter who.)
#define LB_SEEN 512
#ifndef LB_DEDUPE
#define LB_DEDUPE 1
#endif
__device__ __forceinline__ void border_union(int *label, int la, int lb, bool want, unsigned long long *seen = nullptr) {
  if (!want) { la = -1; lb = -2; }
  const int pa = __shfl_up(la, 1), pb = __shfl_up(lb, 1);
  bool mine = want && !(__lane_id() > 0 && pa == la && pb == lb);
  if (mine && seen != nullptr) {
    const unsigned long long key = ((unsigned long long)(unsigned)la << 32) | (unsigned)lb;
    unsigned h = ((unsigned)la * 2654435761u ^ (unsigned)lb * 40503u) >> 23;
    for (int probe = 0; probe < 4; probe++) {
      const unsigned long long was = atomicCAS(&seen[h], ~0ull, key);
      if (was == key) { mine = false; break; }      // somebody of this block is on it
      if (was == ~0ull) break;                       // claimed
      h = (h + 1) & (LB_SEEN - 1);
    }                                                // (a crowded neighbourhood: just do it)
  }
  if (mine) {
    int a = la, b = lb;
    for (;;) {
      // both walks to the roots together: they are chains of dependent loads (the background's tile roots form chains as long as a row of tiles until
      // the shortcuts below shorten them), and a wave waits for the longest - one walk after the other 61 us per launch of 8 frames, together 46.
      // (Not kept: hanging every node that is left under its grandparent on the way, 66 us; a thread's three candidate pairs walked and hooked together, 75 us.)
      int na = ld_agent(label + a), nb = ld_agent(label + b);      // (device-scope loads of a global plane; a `volatile` pointer made them flat, system-scope ones)
      while (na != a || nb != b) { a = na; b = nb; na = ld_agent(label + a); nb = ld_agent(label + b); }
      if (a == b) break;
      if (a < b) { const int t = a; a = b; b = t; }
      const int old = atomicMin(&label[a], b);
      if (old == a) break;
      a = old;
    }
    // shortcut: both tile roots now point (at least) as far as the common root found - later walks through them are short
    const int r = a < b ? a : b;
    if (r < la) atomicMin(&label[la], r);
    if (r < lb) atomicMin(&label[lb], r);
  }
}

// (both kinds of border in one launch - unions commute: the first `hblocks` blocks take the horizontal borders)
__global__ __launch_bounds__(256) void k_label_border(int *label, const int *__restrict__ pix, int bgc, int iw, int ih, int hblocks, size_t zs) {
  RD_ZSHIFT(zs, label, pix);
#if LB_DEDUPE
  __shared__ unsigned long long seen_tab[LB_SEEN];
  for (int i = threadIdx.x; i < LB_SEEN; i += 256) seen_tab[i] = ~0ull;
  __syncthreads();
  unsigned long long *const seen = seen_tab;
#else
  unsigned long long *const seen = nullptr;
#endif
  const bool horizontal = (int)blockIdx.x < hblocks;
  const int t = (horizontal ? blockIdx.x : blockIdx.x - hblocks) * blockDim.x + threadIdx.x;
  if (horizontal) {
    const int nb = (ih - 1) / LT_H;              // border rows
    const int x = t % iw, k = t / iw;
    const bool in = k < nb;
    const int y = (k + 1) * LT_H;
    const int p = y * iw + x;
    const int v = in ? pix[p] : bgc;
    const bool act = in && v != bgc;
    const bool wSame = act && x > 0 && pix[p - 1] == v;
    const bool nSame = act && pix[p - iw] == v;
    const bool nwSame = act && x > 0 && pix[p - iw - 1] == v;
    const bool neSame = act && x < iw - 1 && pix[p - iw + 1] == v;
    const bool eSame = act && x < iw - 1 && pix[p + 1] == v;
    // (the labels of all three candidate pairs are requested together: clamped addresses, used only where wanted)
    const int pc = in ? p : 0;
    const int l0 = label[pc], ln = label[in ? p - iw : 0], lnw = label[(in && x > 0) ? p - iw - 1 : 0], lne = label[(in && x < iw - 1) ? p - iw + 1 : 0];
    border_union(label, l0, ln, nSame && !(wSame && nwSame), seen);
    border_union(label, l0, lnw, !nSame && nwSame && !wSame, seen);
    border_union(label, l0, lne, !nSame && neSame && !eSame, seen);
  } else {
    const int nb = (iw - 1) / LT_W;              // border columns
    const int y = t % ih, k = t / ih;
    const bool in = k < nb;
    const int x = (k + 1) * LT_W;
    const int p = y * iw + x;
    const int v = in ? pix[p] : bgc;
    const bool act = in && v != bgc;
    const bool wSame = act && pix[p - 1] == v;
    const bool nSame = act && y > 0 && pix[p - iw] == v;
    const bool nwSame = act && y > 0 && pix[p - iw - 1] == v;
    const int pc = in ? p : 0;
    const int l0 = label[pc], lw = label[in ? p - 1 : 0], lnw = label[(in && y > 0) ? p - iw - 1 : 0];
    const int ll0 = label[in ? p - 1 : 0], llne = label[(in && y > 0) ? p - 1 - iw + 1 : 0];
    border_union(label, l0, lw, wSame, seen);
    border_union(label, l0, lnw, nwSame && !nSame && !wSame, seen);
    // the pixel left of the border and its NE neighbour (x, y-1)
    const int pl = p - 1;
    const int vl = in ? pix[pl] : bgc;
    const bool actl = in && vl != bgc && y > 0;
    const bool lne = actl && pix[pl - iw + 1] == vl;
    const bool ln = actl && pix[pl - iw] == vl;
    const bool le = actl && pix[pl + 1] == vl;
    border_union(label, ll0, llne, lne && !ln && !le, seen);
  }
}

// Phase 3: path compression to the root
// (vt_*, optional: block 0 also undoes the previous frame's entries of the vote tables - what k_reduce_clean does - so that the
//  vote kernels that follow the boundary labelling find them clean without a launch of their own)
__global__ __launch_bounds__(256) void k_label_flatten(int *label, int n, int *vt_table, int *vt_claim, int *vt_list, size_t zs) {
  RD_ZSHIFT(zs, label, vt_table, vt_claim, vt_list);
  const int stride = gridDim.x * blockDim.x;
  for (int i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += stride * 4) {      // four pixels per step: their first loads are in flight together
    int l[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { const int i = i0 + k * stride; l[k] = i < n ? label[i] : -1; }
    int r[4], n[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { r[k] = l[k] < 0 ? 0 : l[k]; n[k] = l[k] < 0 ? 0 : label[r[k]]; }
    while ((n[0] != r[0]) | (n[1] != r[1]) | (n[2] != r[2]) | (n[3] != r[3])) {      // the four walks take their steps together
#pragma unroll
      for (int k = 0; k < 4; k++) { r[k] = n[k]; const int v = label[r[k]]; n[k] = l[k] < 0 ? 0 : v; }      // (unconditional loads: a walk that has arrived re-reads its root, one that never started reads word 0 and stays at 0)
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (l[k] >= 0 && r[k] != l[k]) label[i0 + k * stride] = r[k];
  }
  if (vt_list != nullptr && blockIdx.x == 0) {
    const int m = vt_list[0];
    for (int j = threadIdx.x; j < m; j += blockDim.x) {
      const int slot = vt_list[1 + j];
      int *e = vt_table + (size_t)slot * 5;
      e[0] = 0; e[1] = 0; e[2] = 0; e[3] = 0; e[4] = 0;
      vt_claim[slot] = 0x7f7f7f7f;
    }
    __syncthreads();
    if (threadIdx.x == 0) vt_list[0] = 0;
  }
}

// oclimgutil.cl:641-649: out[label] += (int)(e*e*10000) for interior pixels with label > 0.  A third of the pixels of a noisy frame
// contribute, to hundreds of different components per block: the block sums per label in an LDS hash (one LDS atomic per
// contributing pixel, no loop over the distinct labels of a wave) and touches each of its labels once in global memory at the end;
// zero contributions (most pixels) are skipped.  Integer addition: order independent.
#define CS_ROWS 4      // rows per thread: their loads are in flight together
#define CS_T 1024      // hash slots per block (64 x 16 pixels, a third of them contributing)
// (flatten: the labels arrive as the trees the border kernel left - phase 3 of the labelling, k_label_flatten, is done here on the way:
//  each pixel walks to its root and stores it; any interleaving only ever stores roots)
__global__ __launch_bounds__(256) void k_calc_strength(int *out, const float *__restrict__ edge, int *label, int iw, int ih, const int8_t *__restrict__ add, int flatten, size_t zs) {
  RD_ZSHIFT(zs, out, edge, label);      // (`add` belongs to no frame slot: group launches pass none)
  __shared__ int keys[CS_T], vals[CS_T];
  const int tid = threadIdx.y * 64 + threadIdx.x;
  for (int t = tid; t < CS_T; t += 256) { keys[t] = -1; vals[t] = 0; }
  const int x = blockIdx.x * 64 + threadIdx.x, yb = blockIdx.y * (4 * CS_ROWS) + threadIdx.y;
  int ls[CS_ROWS], as[CS_ROWS];
  float es[CS_ROWS];
#pragma unroll
  for (int k = 0; k < CS_ROWS; k++) {
    const int y = yb + 4 * k;
    const int p = (x < iw && y < ih) ? y * iw + x : 0;
    ls[k] = label[p];
    es[k] = edge[p];
    as[k] = add != nullptr ? add[p] : 0;
  }
  if (flatten) {
    // the thread's walks to the roots take their steps together (chains of dependent loads: the background's are as long as the border kernel left them)
    int r[CS_ROWS], n[CS_ROWS];
    bool act[CS_ROWS];
#pragma unroll
    for (int k = 0; k < CS_ROWS; k++) { act[k] = x < iw && yb + 4 * k < ih && ls[k] >= 0; r[k] = act[k] ? ls[k] : 0; }
#pragma unroll
    for (int k = 0; k < CS_ROWS; k++) n[k] = act[k] ? label[r[k]] : 0;
    for (;;) {
      bool moving = false;
#pragma unroll
      for (int k = 0; k < CS_ROWS; k++) moving = moving || n[k] != r[k];
      if (!moving) break;
#pragma unroll
      for (int k = 0; k < CS_ROWS; k++) { r[k] = n[k]; const int v = label[r[k]]; n[k] = act[k] ? v : 0; }      // (unconditional loads: a walk that has arrived re-reads its root, one that never started reads word 0 and stays at 0)
    }
#pragma unroll
    for (int k = 0; k < CS_ROWS; k++)
      if (act[k] && r[k] != ls[k]) { label[(yb + 4 * k) * iw + x] = r[k]; ls[k] = r[k]; }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < CS_ROWS; k++) {
    const int y = yb + 4 * k;
    // (quirk H1 without a copy launch: the sums start from last frame's strong mask - `out` arrives zeroed and the mask is added
    //  element by element with the same commutative atomics as the sums)
    if (x < iw && y < ih && as[k] != 0) atomicAdd(&out[y * iw + x], as[k]);
    int l = -1, val = 0;
    if (x > 0 && y > 0 && x < iw - 1 && y < ih - 1) {
      l = ls[k];
      if (l > 0) { const float e = es[k]; val = (int)(e * e * 10000.0f); }
    }
    if (l > 0 && val != 0) {
      unsigned h = ((unsigned)l * 2654435761u) >> 22;
      int probes = 0;
      for (;;) {
        const int kprev = atomicCAS(&keys[h], -1, l);
        if (kprev == -1 || kprev == l) { atomicAdd(&vals[h], val); break; }
        h = (h + 1) & (CS_T - 1);
        if (++probes == 32) { atomicAdd(&out[l], val); break; }      // (table crowded around this slot: straight to memory)
      }
    }
  }
  __syncthreads();
  for (int t = tid; t < CS_T; t += 256) if (keys[t] != -1 && vals[t] != 0) atomicAdd(&out[keys[t]], vals[t]);
}

// oclimgutil.cl:651-657
__global__ __launch_bounds__(256) void k_filter_strength(int *label, const int *__restrict__ str, int thre, int iw, int ih) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x <= 0 || y <= 0 || x >= iw - 1 || y >= ih - 1) return;
  const int p = y * iw + x;
  const int l = label[p];
  if (l <= 0 || str[l] < thre) label[p] = -1;
}

// What the frame path needs of the strength sums in one pass over (label, sum): the strong mask that two k_filter_strength
// passes (thresholds t_edge <= t_strong) followed by `label > 0` would produce (twice: this frame's copy and the plane handed to
// the next frame - all the next frame needs from this one; interior pixels keep a label > 0 exactly when their sum
// reaches t_strong, the frame ring is never filtered), the edge mask at t_edge as int and int8 (oclrect.c:277-284) - both from
// the unfiltered labels - and the labels filtered at t_strong in place (filtering at t_edge first changes nothing).
__device__ __forceinline__ void strength_mask_one(int l, int sum, bool interior, int t_edge, int t_strong, int &vs, int &ve, int &lnew) {
  vs = (l > 0 && !(interior && sum < t_strong)) ? 1 : 0;
  ve = (l > 0 && !(interior && sum < t_edge)) ? 1 : 0;
  lnew = (interior && l != -1 && (l <= 0 || sum < t_strong)) ? -1 : l;
}
// (prev, optional: the strong mask of the frame before - quirk H1 - as a 0/1 byte plane that is added to the sums element by element:
//  sum of label l = str[l] + prev[l]; then `strong2` must be another plane, other threads still read this one)
// VEC: four consecutive pixels of a row per thread (rows are 16-byte aligned: iw % 4 == 0) - one 16-byte load of the labels, the eight
// gathers of the sums in flight together, 16-byte / 4-byte stores instead of four 4-byte / 1-byte ones.  This kernel is the frame-to-frame
// chain: it runs alone, once per frame.
// bits (optional): the strong mask as a bit plane as well - ceil(iw / 64) words per row, bit b of word wx = pixel wx * 64 + b - which is what the
// polyline stage traces (rd_k_poly.hip); strong (the int plane) may then be null
// 16 bits -> every fourth bit of a 64-bit word
__device__ __forceinline__ unsigned long long spread4(unsigned long long x) {
  x = (x | (x << 24)) & 0x000000ff000000ffull;
  x = (x | (x << 12)) & 0x000f000f000f000full;
  x = (x | (x << 6)) & 0x0303030303030303ull;
  x = (x | (x << 3)) & 0x1111111111111111ull;
  return x;
}
template <bool VEC>
__global__ __launch_bounds__(256) void k_strength_masks(int *__restrict__ strong, int8_t *__restrict__ strong2, int *__restrict__ edge, int8_t *__restrict__ edge8,
                                                         int *__restrict__ label, const int *__restrict__ str, int t_edge, int t_strong, int iw, int ih, const int8_t *__restrict__ prev,
                                                         unsigned long long *__restrict__ bits) {
  if (VEC) {
    const int x = (blockIdx.x * 64 + threadIdx.x) * 4, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= iw || y >= ih) return;      // (lanes beyond the row are simply missing from the ballots below: their bits are 0)
    const int p = y * iw + x;
    const int4 lv = *(const int4 *)(label + p);
    const int l[4] = { lv.x, lv.y, lv.z, lv.w };
    const bool rowin = y > 0 && y < ih - 1;
    bool interior[4]; int sa[4], sb[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      interior[k] = rowin && x + k > 0 && x + k < iw - 1;
      const bool need = l[k] > 0 && interior[k];
      sa[k] = str[need ? l[k] : 0];
      sb[k] = prev ? (int)prev[need ? l[k] : 0] : 0;
    }
    int vs[4], ve[4], ln[4];
#pragma unroll
    for (int k = 0; k < 4; k++) strength_mask_one(l[k], (l[k] > 0 && interior[k]) ? sa[k] + sb[k] : 0, interior[k], t_edge, t_strong, vs[k], ve[k], ln[k]);
    if (strong != nullptr) *(int4 *)(strong + p) = make_int4(vs[0], vs[1], vs[2], vs[3]);
    *(uint32_t *)(strong2 + p) = (uint32_t)vs[0] | ((uint32_t)vs[1] << 8) | ((uint32_t)vs[2] << 16) | ((uint32_t)vs[3] << 24);
    if (bits != nullptr) {
      // the wave's 256 pixels = four words: ballot k holds pixel 4 L + k at bit L; lane j < 4 assembles word j from the j-th 16 lanes
      const unsigned long long b0 = __ballot(vs[0] != 0), b1 = __ballot(vs[1] != 0), b2 = __ballot(vs[2] != 0), b3 = __ballot(vs[3] != 0);
      const int j = threadIdx.x;
      const int wx = blockIdx.x * 4 + j, wpr = (iw + 63) >> 6;
      if (j < 4 && wx < wpr) {
        const int sh = 16 * j;
        bits[(size_t)y * wpr + wx] = spread4((b0 >> sh) & 0xffffull) | (spread4((b1 >> sh) & 0xffffull) << 1) | (spread4((b2 >> sh) & 0xffffull) << 2) | (spread4((b3 >> sh) & 0xffffull) << 3);
      }
    }
    if (edge != nullptr) *(int4 *)(edge + p) = make_int4(ve[0], ve[1], ve[2], ve[3]);
    *(uint32_t *)(edge8 + p) = (uint32_t)ve[0] | ((uint32_t)ve[1] << 8) | ((uint32_t)ve[2] << 16) | ((uint32_t)ve[3] << 24);
    if (ln[0] != l[0] || ln[1] != l[1] || ln[2] != l[2] || ln[3] != l[3]) *(int4 *)(label + p) = make_int4(ln[0], ln[1], ln[2], ln[3]);
    return;
  }
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= iw || y >= ih) return;
  const int p = y * iw + x;
  const int l = label[p];
  const bool interior = x > 0 && y > 0 && x < iw - 1 && y < ih - 1;
  const int sum = (l > 0 && interior) ? str[l] + (prev ? (int)prev[l] : 0) : 0;
  int vs, ve, ln;
  strength_mask_one(l, sum, interior, t_edge, t_strong, vs, ve, ln);
  if (bits != nullptr) { const unsigned long long m = __ballot(vs != 0); if (threadIdx.x == 0) bits[(size_t)y * ((iw + 63) >> 6) + blockIdx.x] = m; }
  if (strong != nullptr) strong[p] = vs;
  strong2[p] = (int8_t)vs;       // (the copy for the next frame: a byte plane - it is read once, as an addend)
  if (edge != nullptr) edge[p] = ve;             // (the int form of the edge mask is a test plane only: rd_detector_debug_plane widens the bytes)
  edge8[p] = (int8_t)ve;
  if (ln != l) label[p] = ln;
}

// The strong masks of the nz consecutive frames of a group launch in ONE launch (frame = blockIdx.z, planes zs bytes apart), where the frame-by-frame form above
// needs nz launches one after the other: frame z's sums start from the strong mask of frame z - 1 (H1).  That mask is 0 or 1, so it only decides where a sum stands
// exactly one below a threshold - sum = t - 1 - and then it is needed at ONE pixel, the component's root: the thread evaluates the mask of the frame before at that
// pixel itself (the same rule, one level down, and so on to the frame before the group, whose mask plane - ring plane t0 mod nring - is complete: the launch waits
// for the group before).  A label read from an earlier frame of the group may or may not have been filtered by that frame's own threads yet: a pixel is filtered
// exactly when its mask is 0, so both readings answer the same.  Every frame still leaves its mask in the ring (plane (t0 + z + 1) mod nring) for whoever comes next.
// 16-byte accesses: needs iw % 4 == 0 and aligned planes like the VEC form above.
__device__ int strong_before(const int *label0, const int *str0, const int8_t *ring0, size_t zs, int zz, int q, int iw, int ih, int t_strong) {
  for (;;) {
    if (zz < 0) return (int)ring0[q];
    // (a word another block of this launch may be rewriting right now - filtered or not, it answers the same, see above: a relaxed atomic load of agent scope, so that the
    //  race is a defined one and the load is neither hoisted nor served from a non-coherent path)
    const int l2 = ld_agent((const int *)((const char *)label0 + (size_t)zz * zs + (size_t)q * 4));
    if (l2 <= 0) return 0;
    const int qy = q / iw, qx = q - qy * iw;
    if (!(qx > 0 && qy > 0 && qx < iw - 1 && qy < ih - 1)) return 1;      // (the frame's ring is never filtered)
    const int s2 = *(const int *)((const char *)str0 + (size_t)zz * zs + (size_t)l2 * 4);
    if (s2 >= t_strong) return 1;
    if (s2 + 1 < t_strong) return 0;
    q = l2; zz--;           // the sum stands one below the threshold: the frame before decides, at this component's root
  }
}
// (label and ring without __restrict__: strong_before reads other frames' label words through an alias while blocks of this launch rewrite them)
__global__ __launch_bounds__(256) void k_strength_masks_group(int8_t *ring, int8_t *__restrict__ edge8, int *label, const int *__restrict__ str, int t_edge, int t_strong,
                                                               int iw, int ih, unsigned long long *__restrict__ bits, long t0, int nring, size_t zs) {
  const int z = blockIdx.z;
  const int *label0 = label, *str0 = str;
  const size_t N = (size_t)iw * ih;
  const int8_t *ring0 = ring + (size_t)(t0 % nring) * N;
  int8_t *out = ring + (size_t)((t0 + z + 1) % nring) * N;
  RD_ZSHIFT(zs, edge8, label, str, bits);
  const int x = (blockIdx.x * 64 + threadIdx.x) * 4, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= iw || y >= ih) return;
  const int p = y * iw + x;
  const int4 lv = *(const int4 *)(label + p);
  const int l[4] = { lv.x, lv.y, lv.z, lv.w };
  const bool rowin = y > 0 && y < ih - 1;
  bool interior[4], need[4]; int sa[4], sb[4];
  bool ask = false;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    interior[k] = rowin && x + k > 0 && x + k < iw - 1;
    need[k] = l[k] > 0 && interior[k];
    sa[k] = str[need[k] ? l[k] : 0];
  }
#pragma unroll
  for (int k = 0; k < 4; k++) { sb[k] = 0; ask = ask || (need[k] && (sa[k] == t_strong - 1 || sa[k] == t_edge - 1)); }
  if (__any(ask)) {
#pragma unroll
    for (int k = 0; k < 4; k++) if (need[k] && (sa[k] == t_strong - 1 || sa[k] == t_edge - 1)) sb[k] = strong_before(label0, str0, ring0, zs, z - 1, l[k], iw, ih, t_strong);
  }
  int vs[4], ve[4], ln[4];
#pragma unroll
  for (int k = 0; k < 4; k++) strength_mask_one(l[k], need[k] ? sa[k] + sb[k] : 0, interior[k], t_edge, t_strong, vs[k], ve[k], ln[k]);
  *(uint32_t *)(out + p) = (uint32_t)vs[0] | ((uint32_t)vs[1] << 8) | ((uint32_t)vs[2] << 16) | ((uint32_t)vs[3] << 24);
  {
    const unsigned long long b0 = __ballot(vs[0] != 0), b1 = __ballot(vs[1] != 0), b2 = __ballot(vs[2] != 0), b3 = __ballot(vs[3] != 0);
    const int j = threadIdx.x;
    const int wx = blockIdx.x * 4 + j, wpr = (iw + 63) >> 6;
    if (j < 4 && wx < wpr) {
      const int sh = 16 * j;
      bits[(size_t)y * wpr + wx] = spread4((b0 >> sh) & 0xffffull) | (spread4((b1 >> sh) & 0xffffull) << 1) | (spread4((b2 >> sh) & 0xffffull) << 2) | (spread4((b3 >> sh) & 0xffffull) << 3);
    }
  }
  *(uint32_t *)(edge8 + p) = (uint32_t)ve[0] | ((uint32_t)ve[1] << 8) | ((uint32_t)ve[2] << 16) | ((uint32_t)ve[3] << 24);
  if (ln[0] != l[0] || ln[1] != l[1] || ln[2] != l[2] || ln[3] != l[3]) *(int4 *)(label + p) = make_int4(ln[0], ln[1], ln[2], ln[3]);
}

}  // namespace

namespace rdk {

// 1 if the frames of a group can take strength_masks_group (16-byte rows and planes)
int strength_masks_group_fits(int iw, const void *label, const void *ring, const void *edge8, size_t zs) {
  return (iw & 3) == 0 && (((uintptr_t)label | zs) & 15) == 0 && (((uintptr_t)ring | (uintptr_t)edge8) & 3) == 0;
}
void strength_masks_group(hipStream_t s, int8_t *ring, int8_t *edge8, int *label, const int *str, int t_edge, int t_strong, int iw, int ih, unsigned long long *bits, long t0, int nring, int nz, size_t zs) {
  hipLaunchKernelGGL(k_strength_masks_group, dim3(cdiv(iw, 256), cdiv(ih, 4), nz), block2, 0, s, ring, edge8, label, str, t_edge, t_strong, iw, ih, bits, t0, nring, zs);
}

void label8(hipStream_t s, int *label, const int *pix, int bgc, int iw, int ih, int skip_flatten) {
  const size_t zs = 0;
  hipLaunchKernelGGL((k_label_tile<0, LT_TY>), dim3(rd_tile_blocks(cdiv(iw, LT_W), cdiv(ih, LT_H), 1)), dim3(64, LT_TY), 0, s, label, pix, bgc, iw, ih, (int *)nullptr, (const float *)nullptr, (int *)nullptr, (int *)nullptr, zs, rd_gdim(cdiv(iw, LT_W), cdiv(ih, LT_H), 1));
  const int nh = ((ih - 1) / LT_H) * iw, nv = ((iw - 1) / LT_W) * ih;
  const int hb = cdiv(nh, 256), vb = cdiv(nv, 256);
  if (hb + vb > 0) hipLaunchKernelGGL(k_label_border, dim3(hb + vb), dim3(256), 0, s, label, pix, bgc, iw, ih, hb, zs);
  if (skip_flatten) return;        // (the caller's next kernel walks to the roots itself: calc_strength)
  const int n = iw * ih;
  int g = cdiv(n, 256 * 4);
  hipLaunchKernelGGL(k_label_flatten, dim3(g < 1 ? 1 : g), dim3(256), 0, s, label, n, (int *)nullptr, (int *)nullptr, (int *)nullptr, zs);
}

// rect_tidy(mask0, tidy, nms, zero_plane) + label8(label, tidy, background -1, skip_flatten) with the tidy computed inside the tile kernel
void label8_tidy(hipStream_t s, int *label, int *mask0, int *tidy, const float *nms, int *zero_plane, int iw, int ih, int skip_flatten, int nz, size_t zs) {
  hipLaunchKernelGGL((k_label_tile<2, 4>), dim3(rd_tile_blocks(cdiv(iw, LT_W), cdiv(ih, LT_H), nz)), dim3(64, 4), 0, s, label, (const int *)nullptr, -1, iw, ih, tidy, nms, mask0, zero_plane, zs, rd_gdim(cdiv(iw, LT_W), cdiv(ih, LT_H), nz));
  const int nh = ((ih - 1) / LT_H) * iw, nv = ((iw - 1) / LT_W) * ih;
  const int hb = cdiv(nh, 256), vb = cdiv(nv, 256);
  // (this plane is labelled with its background, one component that spans the frame: uniting the tiles of a row first and the rows
  //  afterwards keeps the trees that the concurrent unions walk shorter than doing both at once - measured 49 against 56 us alone, 2129
  //  against 2098 frames/s at full rate; blocks of 256 rather than 64 threads: +0.35 % at full rate; the boundary labelling, whose
  //  components are small, is better off with one launch: 2130 against 2123)
  if (vb > 0) hipLaunchKernelGGL(k_label_border, dim3(vb, 1, nz), dim3(256), 0, s, label, (const int *)tidy, -1, iw, ih, 0, zs);
  if (hb > 0) hipLaunchKernelGGL(k_label_border, dim3(hb, 1, nz), dim3(256), 0, s, label, (const int *)tidy, -1, iw, ih, hb, zs);
  if (skip_flatten) return;
  const int n = iw * ih;
  int g = cdiv(n, 256 * 4);
  hipLaunchKernelGGL(k_label_flatten, dim3(g < 1 ? 1 : g, 1, nz), dim3(256), 0, s, label, n, (int *)nullptr, (int *)nullptr, (int *)nullptr, zs);
}

// region boundaries (oclrect.cl:373-390) marked into `marks` and their 8-connected components labelled into `label`
void label8_boundary(hipStream_t s, int *label, int *marks, const int *region, int iw, int ih, int *vt_table, int *vt_claim, int *vt_list, int nz, size_t zs, int flatten) {
  hipLaunchKernelGGL((k_label_tile<1, LT_TY1>), dim3(rd_tile_blocks(cdiv(iw, LT_W), cdiv(ih, LT_H), nz)), dim3(64, LT_TY1), 0, s, label, region, -1, iw, ih, marks, (const float *)nullptr, (int *)nullptr, (int *)nullptr, zs, rd_gdim(cdiv(iw, LT_W), cdiv(ih, LT_H), nz));
  const int nh = ((ih - 1) / LT_H) * iw, nv = ((iw - 1) / LT_W) * ih;
  const int hb = cdiv(nh, 256), vb = cdiv(nv, 256);
  if (hb + vb > 0) hipLaunchKernelGGL(k_label_border, dim3(hb + vb, 1, nz), dim3(256), 0, s, label, (const int *)marks, -1, iw, ih, hb, zs);
  // flatten != 0: every pixel walks to its component's root (phase 3).  The frame path leaves the forest as it is - the plane's few readers walk themselves (rd_k_rect.hip:
  // boundary_root) - and then only the vote tables' clean-up, which rode on that launch, is left: one block
  const int n = iw * ih;
  int g = flatten ? cdiv(n, 256 * 4) : 1;
  hipLaunchKernelGGL(k_label_flatten, dim3(g < 1 ? 1 : g, 1, nz), dim3(256), 0, s, label, flatten ? n : 0, vt_table, vt_claim, vt_list, zs);
}
// phase 3 alone, on a plane label8_boundary(..., flatten = 0) left as a forest (the debug plane "boundary")
void label8_flatten(hipStream_t s, int *label, int n) {
  int g = cdiv(n, 256 * 4);
  hipLaunchKernelGGL(k_label_flatten, dim3(g < 1 ? 1 : g), dim3(256), 0, s, label, n, (int *)nullptr, (int *)nullptr, (int *)nullptr, (size_t)0);
}

void calc_strength(hipStream_t s, int *out, const float *edge, int *label, int iw, int ih, const int8_t *add, int flatten, int nz, size_t zs) {
  hipLaunchKernelGGL(k_calc_strength, dim3(cdiv(iw, 64), cdiv(ih, 4 * CS_ROWS), nz), block2, 0, s, out, edge, label, iw, ih, add, flatten, zs);
}

void strength_masks(hipStream_t s, int *strong, int8_t *strong2, int *edge, int8_t *edge8, int *label, const int *str, int t_edge, int t_strong, int iw, int ih, const int8_t *prev, unsigned long long *bits) {
  // (16-byte accesses need 16-byte aligned rows and planes: the frame path's planes are; an operator call with odd pointers takes the plain form)
  const bool vec = (iw & 3) == 0 && ((((uintptr_t)strong | (uintptr_t)label | (uintptr_t)edge) & 15) == 0) && ((((uintptr_t)strong2 | (uintptr_t)edge8) & 3) == 0);
  if (vec) hipLaunchKernelGGL(k_strength_masks<true>, dim3(cdiv(iw, 256), cdiv(ih, 4)), block2, 0, s, strong, strong2, edge, edge8, label, str, t_edge, t_strong, iw, ih, prev, bits);
  else hipLaunchKernelGGL(k_strength_masks<false>, grid2(iw, ih), block2, 0, s, strong, strong2, edge, edge8, label, str, t_edge, t_strong, iw, ih, prev, bits);
}

void filter_strength(hipStream_t s, int *label, const int *str, int thre, int iw, int ih) {
  hipLaunchKernelGGL(k_filter_strength, grid2(iw, ih), block2, 0, s, label, str, thre, iw, ih);
}

}  // namespace rdk
