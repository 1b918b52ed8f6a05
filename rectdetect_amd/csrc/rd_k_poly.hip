// rectdetect-mi355x: polyline extraction for gfx950 (the reference's oclpolyline_execute, oclpolyline.c:218-309,
// kernels oclpolyline.cl = "pl").
//
// The reference runs 116 full-frame launches over dense int planes although only chain pixels (a few per cent of the
// frame at most) take part after the tidy step.  Here the tidy step is dense, then chain pixels are compacted IN RASTER
// ORDER (ballot + prefix scan) and every later step works on the compact arrays; raster-ordered compaction makes the
// reference's order-dependent id hand-outs plain prefix ranks.  Pointer-jumping steps keep the
// reference's hop counts (4 x 8 hops, 3 x 32 hops) so results agree even where those limits bite.
#include "rd_device.h"
#include "rd_kernels.h"

#include "rd_poly_scratch.h"
#include "rd_tidy_tile.h"

namespace {

using namespace rd;
using rdk::PolyScratch;
using rdk::PolyFrame;
// the frame a block works on (see PolyFrame): its descriptor lives in the kernel-argument segment (uniform scalar loads, indexable)
#define RD_FRAME const PolyFrame &FRM = FRS.f[blockIdx.z]; const PolyScratch &s = FRM.ps

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
const dim3 block2(64, 4);
inline dim3 grid2(int iw, int ih) { return dim3(cdiv(iw, 64), cdiv(ih, 4)); }
#define RD_XY const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y
#define SPARSE_GRID 512
#define SPARSE_LOOP(i, cnt) for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (cnt); i += gridDim.x * blockDim.x)

struct ls_rec { float x0, y0, x1, y1; int startIndex, endIndex, leftPtr, rightPtr, startCount, endCount, maxDist, polyid, npix, level; };
struct lsx_rec { long long mx00, mx01, mx11, my0, my1; short dx, dy, vx, vy; int d2, pad; };

// ------------------------------------------------------------------------------------------------ tidy and compaction on bit planes
// The mask whose curves are traced arrives as a BIT PLANE: wpr = ceil(iw / 64) words per row, bit b of word (y, wx) = pixel (wx * 64 + b, y),
// bits beyond the frame 0 (the rect path's strong-mask kernel writes it directly - rd_k_label.hip: k_strength_masks -; a caller's int plane is
// converted by k_mask_bits).  Everything up to the chain pixels' compact arrays then works on 1 bit per pixel:
//   k_tidy_bits      the five stencils of the reference (below) on bit rows, a wave per 64-column x 52-row tile, one ROW per lane (the rows above
//                    and below a lane's row come from its neighbour lanes); result: another bit plane
//   k_row_prefix     per word: the number of chain pixels before it in its row; per row: their total
//   k_chain_scatter  raster-order ranks = row base + word prefix + bits below: pos[rank] = pixel; the row bases are written on the way
// and a pixel's compact index - what the reference's dense planes answer by position - is a lookup in those tables (px_rank).  Before: a dense
// int plane out of the tidy, a dense compaction pass over it, a dense pixel -> rank plane: 16 bytes per pixel and 27 us per 1080p frame of
// sweeps to find 4 % of the pixels.
//
// The five stencils:
//   pl:66-87   junction counts (`!= 0`): on-pixels of the 3x3 block, isolated pixels -> 0, frame border 0
//   pl:89-110  1-px gaps between two curve ends are bridged (8 strict patterns).  The kernel of the reference does not
//              write the 2-px frame ring of its output plane, so the ring keeps the plane's previous content (SURVEY.md
//              H3): here it is taken from ring_src (the caller's plane) or, when that is null, set to ring_const
//   pl:112-124 checkerboard thinning, parity 0 then parity 1
//   pl:126-147 keep on-pixels with at most two on-neighbours (cuts the curves at junctions)
// (6 cells of margin in total; after the bridging step only "zero / non-zero" matters, so every intermediate is a bit)
typedef unsigned long long u64;
#define PT_M 6
#define TB_OUT (64 - 2 * PT_M)          // rows a wave produces

// int plane(s) -> bit planes (callers of oclpolyline_execute hand over int planes): one wave per row and word
__global__ __launch_bounds__(256) void k_mask_bits(const rdk::PolyFrames FRS, int iw, int ih, int wpr) {
  RD_FRAME;
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (y >= ih) return;
  const bool in = x < iw;
  const int v = FRM.in[in ? y * iw + x : 0];
  const u64 m = __ballot(in && v != 0);
  u64 q = 0;
  if (FRM.ring_src != nullptr) {      // (the ring only: rows 0, 1, ih-2, ih-1 and columns 0, 1, iw-2, iw-1)
    const bool ring = in && (y <= 1 || y >= ih - 2 || x <= 1 || x >= iw - 2);
    const int g = ring ? FRM.ring_src[y * iw + x] : 0;
    q = __ballot(g != 0);
  }
  if (threadIdx.x == 0) { s.sb[y * wpr + blockIdx.x] = m; s.rb[y * wpr + blockIdx.x] = q; }
}

__device__ __forceinline__ bitrow br_lane(bitrow a, int src_lane) {
  const unsigned a0 = (unsigned)a.lo, a1 = (unsigned)(a.lo >> 32), a2 = (unsigned)a.hi, a3 = (unsigned)(a.hi >> 32);
  const unsigned b0 = __shfl(a0, src_lane), b1 = __shfl(a1, src_lane), b2 = __shfl(a2, src_lane), b3 = __shfl(a3, src_lane);
  return br((u64)b0 | ((u64)b1 << 32), (u64)b2 | ((u64)b3 << 32));
}

__global__ __launch_bounds__(256) void k_tidy_bits(const rdk::PolyFrames FRS, int ring_const, int iw, int ih, int wpr) {
  RD_FRAME;
  const u64 *__restrict__ SB = FRM.in_bits != nullptr ? FRM.in_bits : s.sb;
  const u64 *__restrict__ RB = (FRM.in_bits == nullptr && FRM.ring_src != nullptr) ? s.rb : nullptr;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) s.csync[0] = s.csync[0] + 1;      // generation of this frame's compaction state words (k_compact1)
  const int lane = threadIdx.x & 63, wx = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wx >= wpr) return;                               // (whole waves: the kernel has no block-wide barrier)
  constexpr int R = 64;
  const int x0 = wx * 64, y0 = blockIdx.y * TB_OUT;
  const int r = lane, y = y0 - PT_M + r;
  const bool yin = y >= 0 && y < ih;
  // the lane's row with 6 cells of margin as 128 bits: bit b = column x0 - 6 + b
  bitrow M = br(0, 0), RG = br(0, 0);
  if (yin) {
    const u64 *row = SB + (size_t)y * wpr + wx;
    const u64 W = wx > 0 ? row[-1] : 0ull, C = row[0], E = wx + 1 < wpr ? row[1] : 0ull;
    M = br((W >> 58) | (C << 6), (C >> 58) | (E << 6));
    if (RB != nullptr) {
      const u64 *rr = RB + (size_t)y * wpr + wx;
      const u64 W2 = wx > 0 ? rr[-1] : 0ull, C2 = rr[0], E2w = wx + 1 < wpr ? rr[1] : 0ull;
      RG = br((W2 >> 58) | (C2 << 6), (C2 >> 58) | (E2w << 6));
    }
  }
  const int b0 = PT_M - x0;                            // bit of column 0
  const bitrow inimg = yin ? br_range(b0, iw - 1 + b0) : br(0, 0);
  const bitrow in1 = (y >= 1 && y <= ih - 2) ? br_range(1 + b0, iw - 2 + b0) : br(0, 0);
  const bitrow core = (y >= 2 && y <= ih - 3) ? br_range(2 + b0, iw - 3 + b0) : br(0, 0);      // the frame without its 2-px ring
  const u64 even = (((x0 - PT_M + y) & 1) == 0) ? 0x5555555555555555ull : 0xaaaaaaaaaaaaaaaaull;
  const int up = lane > 0 ? lane - 1 : 0, dn = lane < 63 ? lane + 1 : 63, up2 = lane > 1 ? lane - 2 : 0, dn2 = lane < 62 ? lane + 2 : 63;
  // (a lane whose row lies outside a stencil's valid range holds zeros for it, like the rows of the tile kernel this replaces: what it
  //  receives from beyond the wave's ends never matters)
  bitrow NZ = br(0, 0), E2 = br(0, 0);
  {   // pl:66-87: on-pixels with at least one on-neighbour keep a count (!= 0); count 2 = exactly one neighbour
    const bitrow Mu = br_lane(M, up), Md = br_lane(M, dn);
    if (r >= 1 && r < R - 1) {
      bitrow ge1, ge2;
      br_count8(Mu, M, Md, ge1, ge2);
      const bitrow on = M & in1;
      NZ = on & ge1;
      E2 = on & ge1 & ~ge2;
    }
  }
  bitrow O = br(0, 0);
  {   // pl:89-110: on-pixels stay, 1-px gaps between two curve ends are bridged (8 strict patterns); the ring keeps stale values (H3)
    const bitrow z2n = br_lane(NZ, up2), zm = NZ, zs = br_lane(NZ, dn), z2s = br_lane(NZ, dn2), en = br_lane(E2, up), em = E2, es = br_lane(E2, dn);
    if (r >= 3 && r < R - 3) {
      const bitrow wem = br_west(em), eem = br_east(em), w2zm = br_west(br_west(zm)), e2zm = br_east(br_east(zm));
      const bitrow pat = (w2zm & wem & eem & e2zm) |
                         (z2n & en & es & z2s) |
                         (br_west(br_west(z2n)) & br_west(en) & br_east(es) & br_east(br_east(z2s))) |
                         (br_east(br_east(z2n)) & br_east(en) & br_west(es) & br_west(br_west(z2s))) |
                         (e2zm & eem & br_west(es) & br_west(br_west(zs))) |
                         (w2zm & wem & br_east(es) & br_east(br_east(zs))) |
                         (br_east(z2n) & br_east(en) & es & z2s) |
                         (br_west(z2n) & br_west(en) & es & z2s);
      const bitrow ring = inimg & ~core;
      const bitrow ringval = RB != nullptr ? RG : (ring_const != 0 ? br(~0ull, ~0ull) : br(0, 0));
      O = (ring & ringval) | (core & (zm | pat));
    }
  }
  bitrow T0 = br(0, 0);
  {   // pl:112-124, parity 0
    const bitrow Ou = br_lane(O, up), Od = br_lane(O, dn);
    if (r >= 4 && r < R - 4) T0 = O & ~(in1 & br(even, even) & (Ou | Od) & (br_west(O) | br_east(O)));
  }
  bitrow T1 = br(0, 0);
  {   // parity 1
    const bitrow Tu = br_lane(T0, up), Td = br_lane(T0, dn);
    if (r >= 5 && r < R - 5) T1 = T0 & ~(in1 & br(~even, ~even) & (Tu | Td) & (br_west(T0) | br_east(T0)));
  }
  {   // pl:126-147: keep on-pixels with at most two on-neighbours (cuts the curves at junctions)
    const bitrow u = br_lane(T1, up), m = T1, d = br_lane(T1, dn);
    if (r >= PT_M && r < R - PT_M && y < ih) {
      const bitrow nb[8] = { br_west(u), u, br_east(u), br_west(m), br_east(m), br_west(d), d, br_east(d) };
      bitrow ge1 = br(0, 0), ge2 = br(0, 0), ge3 = br(0, 0);
#pragma unroll
      for (int k = 0; k < 8; k++) { ge3 = ge3 | (ge2 & nb[k]); ge2 = ge2 | (ge1 & nb[k]); ge1 = ge1 | nb[k]; }
      const bitrow f = m & in1 & ~ge3;
      s.tb[(size_t)y * wpr + wx] = (f.lo >> PT_M) | (f.hi << (64 - PT_M));
    }
  }
}

// chain pixels before each word in its row (pw), per row (rowsum): a wave per row
__global__ __launch_bounds__(256) void k_row_prefix(const rdk::PolyFrames FRS, int ih, int wpr) {
  RD_FRAME;
  const int lane = threadIdx.x & 63, y = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (y >= ih) return;
  int run = 0;
  for (int w0 = 0; w0 < wpr; w0 += 64) {
    const int i = w0 + lane;
    const int c = i < wpr ? __popcll(s.tb[(size_t)y * wpr + i]) : 0;
    int inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (i < wpr) s.pw[(size_t)y * wpr + i] = run + inc - c;
    run += __shfl(inc, 63);
  }
  if (lane == 0) s.rowsum[y] = run;
}

// compact index of pixel (qx, qy) - its rank among the chain pixels in raster order - or -1 (k_chain_scatter has written the row bases)
__device__ __forceinline__ int px_rank(const PolyScratch &s, int wpr, int qx, int qy) {
  const size_t wi = (size_t)qy * wpr + (qx >> 6);
  const u64 w = s.tb[wi];
  const int b = qx & 63;
  if (!((w >> b) & 1ull)) return -1;
  return s.rowbase[qy] + s.pw[wi] + __popcll(w & ((1ull << b) - 1ull));
}

// pos[rank] = pixel for every chain pixel (raster order), lab[rank] = rank, ends[rank] = 0; rowbase[y] = chain pixels in rows above y; ctr[0] = their number
#define CS_ROWS_PER_BLOCK 8
__global__ __launch_bounds__(256) void k_chain_scatter(const rdk::PolyFrames FRS, int iw, int ih, int wpr) {
  RD_FRAME;
  __shared__ int part[256];
  __shared__ int base[CS_ROWS_PER_BLOCK + 1];
  const int tid = threadIdx.x, ya = blockIdx.x * CS_ROWS_PER_BLOCK;
  int acc = 0;
  for (int i = tid; i < ya; i += 256) acc += s.rowsum[i];
  part[tid] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) part[tid] += part[tid + o]; __syncthreads(); }
  if (tid == 0) {
    int run = part[0];
    for (int j = 0; j < CS_ROWS_PER_BLOCK; j++) {
      base[j] = run;
      if (ya + j < ih) { s.rowbase[ya + j] = run; run += s.rowsum[ya + j]; }
    }
    base[CS_ROWS_PER_BLOCK] = run;
    if (ya + CS_ROWS_PER_BLOCK >= ih) s.ctr[0] = run;      // the block with the frame's last rows
  }
  __syncthreads();
  for (int t = tid; t < CS_ROWS_PER_BLOCK * wpr; t += 256) {
    const int j = t / wpr, wx = t - j * wpr, y = ya + j;
    if (y >= ih) break;
    u64 w = s.tb[(size_t)y * wpr + wx];
    int rank = base[j] + s.pw[(size_t)y * wpr + wx];
    const int p = y * iw + wx * 64;
    while (w) {
      const int b = __ffsll((long long)w) - 1;
      w &= w - 1;
      s.pos[rank] = p + b; s.lab[rank] = rank; s.ends[rank] = 0;
      rank++;
    }
  }
}

// ------------------------------------------------------------------------------------------------ raster-order compaction
#define CP_PER_BLOCK 8192
// Stable (index-ordered) compaction of the non-zero elements of `plane` in ONE launch (chained scan with decoupled look-back):
// a block counts the non-zero elements among its CP_PER_BLOCK, publishes the count, adds up the published counts / running
// totals of the blocks before it (one wave looks at 64 predecessors per step; workgroups are dispatched in index order and a
// launch has so few of them - CP_PER_BLOCK elements each - that the launches of all hardware queues together cannot fill a
// die with waiting blocks: the blocks waited for get to run) and publishes its own running total; then it scatters.  Blocks beyond the element count -
// which may live on the device (nptr) - leave at once.  The state words carry a generation number (*gen, advanced once per
// frame by k_tidy_bits; every call site has its own state array), so nothing has to be cleared between launches.  Outputs (each
// optional): pos[rank] = index, rank1[index] = rank + 1 for non-zero elements; *cnt = their number.
#define CP_K (CP_PER_BLOCK / 256)
__device__ __forceinline__ unsigned long long cp_word(unsigned gen, unsigned status, unsigned value) { return ((unsigned long long)(gen & 0xffffffu) << 40) | ((unsigned long long)status << 38) | value; }
// (MODE 0 - the chain pixels themselves out of a dense plane - is gone: k_chain_scatter.)  MODE 1: element i is "chain pixel i is the root of a sub-chain of more than size_thre pixels"
// (pl:380-420: surviving roots are numbered 1..K in raster order = compact order).  MODE 2: element i is the id of chain pixel i's
// sub-chain (0 = dropped), computed and stored on the way; block 0 also resets what the single-launch stage expects cleared
// (header record, counters ctr[2..23], ctr[25]) when ls != nullptr.
template <int MODE>
__global__ __launch_bounds__(256) void k_compact1(const rdk::PolyFrames FRS, int n, int nblk, int size_thre, int init_ls) {
  RD_FRAME;
  // what the three call sites compact, and where the results go (each has its own state words)
  static_assert(MODE == 1 || MODE == 2, "two call sites");
  int *__restrict__ pos = MODE == 2 ? s.live : nullptr;
  int *__restrict__ rank1 = MODE == 1 ? s.rootid : nullptr;
  const int *nptr = s.ctr;
  int *cnt = MODE == 1 ? s.ctr + 1 : s.ctr + 24;
  unsigned long long *state = s.cstate + (size_t)MODE * nblk;
  const int *genp = s.csync;
  ls_rec *ls = (MODE == 2 && init_ls) ? (ls_rec *)FRM.lslist : nullptr;
  if (MODE == 2 && ls != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
    ls_rec z = {};
    ls[0] = z;
    s.segaux[0] = -1; s.segaux[1] = 0x7fffffff;
    for (int k = 2; k < 24; k++) s.ctr[k] = 0;
    s.ctr[25] = 0;
  }
  __shared__ int wcount[CP_K * 4 + 1];
  __shared__ int s_excl;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int b = blockIdx.x;
  if (nptr) n = *nptr;
  if (b * CP_PER_BLOCK >= n) { if (b == 0 && tid == 0 && cnt) *cnt = 0; return; }
  const unsigned gen = (unsigned)*genp;
  // flags of this thread's CP_K elements (all loads in flight together), wave counts per row of 256 elements
  unsigned on = 0;
  if (MODE == 1) {
    int l[CP_K], sz[CP_K];
#pragma unroll
    for (int k = 0; k < CP_K; k++) { const int i = b * CP_PER_BLOCK + k * 256 + tid; const int q = i < n ? i : 0; l[k] = s.lab2[q]; sz[k] = s.size[q]; }
#pragma unroll
    for (int k = 0; k < CP_K; k++) { const int i = b * CP_PER_BLOCK + k * 256 + tid; if (i < n && l[k] == i && sz[k] > size_thre) on |= 1u << k; }
  } else {
    int l[CP_K];
#pragma unroll
    for (int k = 0; k < CP_K; k++) { const int i = b * CP_PER_BLOCK + k * 256 + tid; l[k] = s.lab2[i < n ? i : 0]; }
#pragma unroll
    for (int k = 0; k < CP_K; k++) l[k] = l[k] >= 0 ? s.rootid[l[k]] : 0;      // ids of the surviving chains for every chain pixel (0 = dropped)
#pragma unroll
    for (int k = 0; k < CP_K; k++) { const int i = b * CP_PER_BLOCK + k * 256 + tid; if (i < n) { s.id[i] = l[k]; if (l[k] != 0) on |= 1u << k; } }
  }
#pragma unroll
  for (int k = 0; k < CP_K; k++) { const unsigned long long m = __ballot((on >> k) & 1u); if (lane == 0) wcount[k * 4 + w] = __popcll(m); }
  __syncthreads();
  if (w == 0) {
    // exclusive scan of the CP_K * 4 wave counts (element order: row k, then wave, then lane); CP_E consecutive entries per lane
    constexpr int CP_E = (CP_K * 4 + 63) / 64;
    int e[CP_E], c = 0;
#pragma unroll
    for (int q = 0; q < CP_E; q++) { e[q] = lane * CP_E + q < CP_K * 4 ? wcount[lane * CP_E + q] : 0; c += e[q]; }
    int inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    int run = inc - c;
#pragma unroll
    for (int q = 0; q < CP_E; q++) { if (lane * CP_E + q < CP_K * 4) wcount[lane * CP_E + q] = run; run += e[q]; }
    const int total = __shfl(inc, 63);
    if (lane == 0) __hip_atomic_store(&state[b], cp_word(gen, b == 0 ? 2u : 1u, (unsigned)total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // look-back
    int excl = 0;
    for (int j = b - 1; j >= 0; j -= 64) {
      const int idx = j - lane;
      unsigned long long st;
      bool ready;
      int spins = 0;
      do {
        st = idx >= 0 ? __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : cp_word(gen, 2u, 0u);
        ready = (unsigned)(st >> 40) == (gen & 0xffffffu) && ((st >> 38) & 3ull) != 0ull;
        if (++spins > (1 << 24)) abort();       // (seconds: an earlier block never came - fail loudly rather than hang)
      } while (!__all(ready));
      const unsigned long long pm = __ballot(((st >> 38) & 3ull) == 2ull);      // predecessors whose running total is known
      const int upto = pm ? __ffsll((long long)pm) - 1 : 63;
      int v = lane <= upto ? (int)(st & 0x3fffffffffull) : 0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      excl += v;
      if (pm) break;
    }
    if (lane == 0) {
      s_excl = excl;
      if (b > 0) __hip_atomic_store(&state[b], cp_word(gen, 2u, (unsigned)(excl + total)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((b + 1) * CP_PER_BLOCK >= n && cnt) *cnt = excl + total;      // the last block with elements
    }
  }
  __syncthreads();
  const int excl = s_excl;
#pragma unroll
  for (int k = 0; k < CP_K; k++) {
    const int i = b * CP_PER_BLOCK + k * 256 + tid;
    const bool o = (on >> k) & 1u;
    const unsigned long long m = __ballot(o);
    if (i >= n) continue;
    const int rank = excl + wcount[k * 4 + w] + __popcll(m & ((1ull << lane) - 1ull));
    if (o) { if (pos) pos[rank] = i; if (rank1) rank1[i] = rank + 1; }
  }
}

// ------------------------------------------------------------------------------------------------ chain graph
// The 8 neighbour compact indices of every chain pixel (E,NE,N,NW,W,SW,S,SE; -1 = none), its degree, and - in the same launch -
// the 8-connected components of the chain mask (pl:811-854 to convergence): union with every neighbour of smaller index.
// (lab[i] = i and ends[i] = 0 were set for all chain pixels by k_chain_scatter)
__global__ void k_chain_union(const rdk::PolyFrames FRS, int iw, int wpr) {
  RD_FRAME;
  const int cnt = s.ctr[0];
  SPARSE_LOOP(i, cnt) {
    const int p = s.pos[i];
    const int py = p / iw, px = p - py * iw;
    int nb[8], deg = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) nb[k] = px_rank(s, wpr, px + nbr_dx(k), py + nbr_dy(k));   // chain pixels are interior: no bounds check needed
#pragma unroll
    for (int k = 0; k < 8; k++) { s.nbr[i * 8 + k] = nb[k]; deg += nb[k] >= 0; }
    s.flag2[i] = deg;     // degree, used for the end-point count
#pragma unroll
    for (int k = 0; k < 8; k++) if (nb[k] >= 0 && nb[k] < i) uf_union(s.lab, i, nb[k]);
  }
}

// Optional work on the way, with the root each pixel has just found (saves a launch each):
//   ends / deg: pl:149-155 - a pixel with junction count 2 (itself + one neighbour, deg == 1) is a chain end; count the ends per chain;
//   size: pl:357-378 - pixels per sub-chain.  The lanes of a wave are consecutive chain pixels in raster order: a run of equal roots
//         (a horizontal stretch of one chain) is counted by its first lane - same-address atomics are served one after the other,
//         a long chain was thousands of them.
// which 0: chain labels (+ ends per chain); 1: sub-chain labels (+ sizes)
__global__ void k_flatten(const rdk::PolyFrames FRS, int which) {
  RD_FRAME;
  int *lab = which == 0 ? s.lab : s.lab2;
  int *ends = which == 0 ? s.ends : nullptr;
  const int *deg = s.flag2;
  int *size = which == 0 ? nullptr : s.size;
  const int cnt = s.ctr[0];
  for (int i0 = (blockIdx.x * blockDim.x + threadIdx.x) & ~63; i0 < cnt; i0 += gridDim.x * blockDim.x) {     // whole waves together
    const int lane = threadIdx.x & 63, i = i0 + lane;
    int r = -1;
    if (i < cnt) {
      const int l = lab[i];
      r = l;
      if (l >= 0) { r = uf_find(lab, l); if (r != l) lab[i] = r; }
      if (ends != nullptr && deg[i] == 1) atomicAdd(&ends[r], 1);
    }
    if (size != nullptr) {
      const int prev = __shfl_up(r, 1);
      const bool start = lane == 0 || prev != r;
      const unsigned long long after = __ballot(start) & ~((2ull << lane) - 1ull);
      if (start && r >= 0) atomicAdd(&size[r], (after ? __ffsll((long long)after) - 1 : 64) - lane);
    }
  }
}

// pl:157-167: a chain without ends is a closed loop: it is opened by deleting its root pixel.
// pl:169-220: the first two living neighbours (order E,NE,N,NW,W,SW,S,SE) become next / prev; self if missing.
// (one launch: whether a pixel is such a root is read off the labels and end counts directly; the pixel's own verdict is stored
//  in alive[] for the launches that follow)
__global__ void k_find_ends0(const rdk::PolyFrames FRS) {
  RD_FRAME;
  const int cnt = s.ctr[0];
  SPARSE_LOOP(i, cnt) {
    // (the eight neighbours, then their labels and end counts: two levels of loads, each issued together)
    int nb[8], lb[8], eb[8];
#pragma unroll
    for (int k = 0; k < 8; k++) nb[k] = s.nbr[i * 8 + k];
    const int li = s.lab[i], ei = s.ends[i];
#pragma unroll
    for (int k = 0; k < 8; k++) { const int j = nb[k] >= 0 ? nb[k] : i; lb[k] = s.lab[j]; eb[k] = s.ends[j]; }
    int a = i, b = i;
    const bool me = !(li == i && ei == 0);
    s.alive[i] = me ? 1 : 0;
    if (me) {
      int found = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const bool living = nb[k] >= 0 && !(lb[k] == nb[k] && eb[k] == 0);
        if (living && found == 0) { a = nb[k]; found = 1; }
        else if (living && found == 1) { b = nb[k]; found = 2; }
      }
    }
    s.nx[0][i] = a; s.pv[0][i] = b;
  }
}

__global__ void k_find_ends0_flags(const rdk::PolyFrames FRS) {
  RD_FRAME;
  const int cnt = s.ctr[0];
  SPARSE_LOOP(i, cnt) {
    int f = 0;
    if (s.alive[i]) {
      const int a = s.nx[0][i], b = s.pv[0][i];
      if (a != i && s.nx[0][a] == i) f |= 1;
      if (b != i && s.pv[0][b] == i) f |= 2;
    }
    s.flag[i] = f;
  }
}

// pl:222-267: eight hops towards both chain ends with orientation-reversal tracking; page selects the flag bit pair
// (final: the last of the four launches also does pl:269-285 for its pixel - link towards the end with the smaller index, the
//  end itself gets number 0 - with the two ends it has just found)
__global__ void k_find_ends1(const rdk::PolyFrames FRS, int page, int final) {
  RD_FRAME;
  const int cnt = s.ctr[0];
  const int *ni = s.nx[page], *pi = s.pv[page];
  int *no = s.nx[page ^ 1], *po = s.pv[page ^ 1];
  const int *fin = page == 0 ? s.flag : s.flag2;
  int *fout = page == 0 ? s.flag2 : s.flag;
  SPARSE_LOOP(i, cnt) {
    if (!s.alive[i]) { no[i] = i; po[i] = i; fout[i] = fin[i]; if (final) { s.num[0][i] = 0; s.link[0][i] = -1; } continue; }
    const int f0 = fin[i];
    bool revn = page == 0 ? (f0 & 1) != 0 : (f0 & 4) != 0;
    bool revp = page == 0 ? (f0 & 2) != 0 : (f0 & 8) != 0;
    int nn = ni[i], pp = pi[i];
    for (int h = 0; h < 8; h++) {
      const int nn2 = revn ? pi[nn] : ni[nn];
      const int pp2 = revp ? ni[pp] : pi[pp];
      int nf = fin[nn], pf = fin[pp];
      if (page != 0) { nf >>= 2; pf >>= 2; }
      revn = revn ? ((nf & 2) == 0) : ((nf & 1) != 0);
      revp = revp ? ((pf & 1) == 0) : ((pf & 2) != 0);
      nn = nn2; pp = pp2;
    }
    no[i] = nn; po[i] = pp;
    int f = f0;
    if (page == 0) { f &= 3; f |= revn ? 4 : 0; f |= revp ? 8 : 0; }
    else { f &= (3 << 2); f |= revn ? 1 : 0; f |= revp ? 2 : 0; }
    fout[i] = f;
    if (final) {
      int nb[8], al[8];
#pragma unroll
      for (int k = 0; k < 8; k++) nb[k] = s.nbr[i * 8 + k];
#pragma unroll
      for (int k = 0; k < 8; k++) al[k] = s.alive[nb[k] >= 0 ? nb[k] : i];
      int a = i, b = i, found = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const bool living = nb[k] >= 0 && al[k] != 0;
        if (living && found == 0) { a = nb[k]; found = 1; }
        else if (living && found == 1) { b = nb[k]; found = 2; }
      }
      const int lk = nn < pp ? a : b;
      s.num[0][i] = lk == i ? 0 : 1; s.link[0][i] = lk;
    }
  }
}

// pl:287-310: 32-hop pointer-jumping prefix sum of the hop counts
// (init_sub: the last round also sets up the sub-chain labelling that follows - pl:312-355 - from the final numbers)
__global__ void k_number(const rdk::PolyFrames FRS, int src, int init_sub) {
  RD_FRAME;
  const int cnt = s.ctr[0];
  const int *ni = s.num[src], *li = s.link[src];
  int *no = s.num[src ^ 1], *lo_ = s.link[src ^ 1];
  SPARSE_LOOP(i, cnt) {
    int n = ni[i], l = li[i];
    if (l != -1) {
      bool ok = true;
      for (int h = 0; h < 32; h++) {
        if (l < 0) { ok = false; break; }
        n += ni[l];
        l = li[l];
      }
      n = ok ? n : 0;
      l = ok ? l : -1;
    }
    no[i] = n;
    lo_[i] = l;
    if (init_sub) { s.lab2[i] = n == 0 ? -1 : i; s.size[i] = 0; s.rootid[i] = 0; }
  }
}

// pl:312-355 to convergence: split chains where the numbering jumps by more than one (set up by the last k_number launch)
__global__ void k_sub_union(const rdk::PolyFrames FRS) {
  RD_FRAME;
  const int *__restrict__ number = s.num[1];
  const int cnt = s.ctr[0];
  SPARSE_LOOP(i, cnt) {
    const int a = number[i];
    if (a == 0) continue;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int j = s.nbr[i * 8 + k];
      if (j < 0 || j > i) continue;
      const int b = number[j];
      if (b == 0) continue;
      const int d = a > b ? a - b : b - a;
      if (d <= 1) uf_union(s.lab2, i, j);
    }
  }
}

// ------------------------------------------------------------------------------------------------ initial segments (pl:439-506)
#define FITS(g, bytes) ((g) >= 0 && (long long)(bytes) > (long long)((g) + 1) * 56ll)

__device__ __forceinline__ void d_seg_clear(const rdk::PolyFrames &FRS, int lsbytes) {
  RD_FRAME;
  ls_rec *ls = (ls_rec *)FRM.lslist;
  const int K = s.ctr[1];
  const int maxrec = lsbytes / 56;
  SPARSE_LOOP(g, K + 1) {
    if (g >= maxrec) continue;
    ls_rec z = {};
    ls[g] = z;
    s.segaux[2 * g] = -1;            // last pixel with number 1
    s.segaux[2 * g + 1] = 0x7fffffff; // first pixel carrying the largest number
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { for (int k = 2; k < 24; k++) s.ctr[k] = 0; s.ctr[25] = 0; }
}
__global__ void k_seg_clear(const rdk::PolyFrames FRS, int lsbytes) { d_seg_clear(FRS, lsbytes); }

__device__ __forceinline__ void d_seg_pass0a(const rdk::PolyFrames &FRS, int lsbytes) {
  RD_FRAME;
  ls_rec *ls = (ls_rec *)FRM.lslist;
  const int *__restrict__ number = s.num[1];
  const int nlive = s.ctr[24];
  SPARSE_LOOP(j, nlive) {
    const int i = s.live[j];
    const int g = s.id[i];
    if (g == 0 || !FITS(g, lsbytes)) continue;
    const int n = number[i];
    if (n == 1) { atomicAdd(&ls[g].startCount, 1); atomicMax(&s.segaux[2 * g], i); }
    atomicAdd(&ls[g].npix, 1);
    atomicMax(&ls[g].endIndex, n);
  }
}
__global__ void k_seg_pass0a(const rdk::PolyFrames FRS, int lsbytes) { d_seg_pass0a(FRS, lsbytes); }

__device__ __forceinline__ void d_seg_pass0b(const rdk::PolyFrames &FRS, int lsbytes) {
  RD_FRAME;
  ls_rec *ls = (ls_rec *)FRM.lslist;
  const int *__restrict__ number = s.num[1];
  const int nlive = s.ctr[24];
  SPARSE_LOOP(j, nlive) {
    const int i = s.live[j];
    const int g = s.id[i];
    if (g == 0 || !FITS(g, lsbytes)) continue;
    if (number[i] != ls[g].endIndex) continue;
    if (ls[g].startCount == 1 && ls[g].npix >= 2) { atomicAdd(&ls[g].endCount, 1); atomicMin(&s.segaux[2 * g + 1], i); }
  }
}
__global__ void k_seg_pass0b(const rdk::PolyFrames FRS, int lsbytes) { d_seg_pass0b(FRS, lsbytes); }

__device__ __forceinline__ void d_seg_finish(const rdk::PolyFrames &FRS, int lsbytes, int iw) {
  RD_FRAME;
  ls_rec *ls = (ls_rec *)FRM.lslist;
  const int K = s.ctr[1];
  SPARSE_LOOP(g0, K) {
    const int g = g0 + 1;
    if (!FITS(g, lsbytes)) continue;
    const int sp = s.segaux[2 * g], ep = s.segaux[2 * g + 1];
    if (sp >= 0) { const int p = s.pos[sp]; ls[g].x0 = (float)(p % iw); ls[g].y0 = (float)(p / iw); }
    if (ls[g].startCount == 1 && ls[g].npix >= 2 && ep != 0x7fffffff) {
      const int p = s.pos[ep];
      ls[g].x1 = (float)(p % iw); ls[g].y1 = (float)(p / iw);
      ls[g].polyid = g;
    } else ls[g].polyid = 0;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { const int maxrec = lsbytes / 56; *(int *)ls = K < maxrec - 1 ? K : (maxrec >= 2 ? maxrec - 2 : 0); }
}
__global__ void k_seg_finish(const rdk::PolyFrames FRS, int lsbytes, int iw) { d_seg_finish(FRS, lsbytes, iw); }

// ------------------------------------------------------------------------------------------------ subdivision rounds (pl:509-646)
__device__ __forceinline__ float dist2f(float vx, float vy, float wx, float wy) { return (vx - wx) * (vx - wx) + (vy - wy) * (vy - wy); }

// pass 3 of the previous round (pixels beyond the new end move right) fused with pass 1 of this round (distance to
// the chord with integer-truncated end points, tie-breaking hash, per-segment maximum)
__device__ __forceinline__ void d_split_move_dist(const rdk::PolyFrames &FRS, int lsbytes, int iw, int do_dist) {
  RD_FRAME;
  ls_rec *ls = (ls_rec *)FRM.lslist;
  const int *__restrict__ number = s.num[1];
  const int nlive = s.ctr[24];
  SPARSE_LOOP(j, nlive) {
    const int i = s.live[j];
    int g = s.id[i];
    if (g == 0 || !FITS(g, lsbytes) || ls[g].polyid == 0) continue;
    if (ls[g].endIndex < number[i]) { g = ls[g].rightPtr; s.id[i] = g; }
    if (!do_dist) continue;
    if (g == 0 || !FITS(g, lsbytes) || ls[g].polyid == 0) continue;
    const int p = s.pos[i], x = p % iw, y = p / iw;
    const float vx = (float)(int)ls[g].x0, vy = (float)(int)ls[g].y0, wx = (float)(int)ls[g].x1, wy = (float)(int)ls[g].y1;
    float cx, cy;
    const float l2 = dist2f(vx, vy, wx, wy);
    if (l2 <= 1e-4f) { cx = vx; cy = vy; }
    else {
      const float t = (((float)x - vx) * (wx - vx) + ((float)y - vy) * (wy - vy)) / l2;
      if (t < 0.0f) { cx = vx; cy = vy; }
      else if (t > 1.0f) { cx = wx; cy = wy; }
      else { cx = vx + t * (wx - vx); cy = vy + t * (wy - vy); }
    }
    const float a = cx - (float)x, b = cy - (float)y;
    int d = (int)((float)sqrt((double)a * (double)a + (double)b * (double)b) * 65536);
    d ^= pixel_rand(p, 0) & 0x1fff;
    s.dist[i] = d;
    atomicMax(&ls[g].maxDist, d);
  }
}
__global__ void k_split_move_dist(const rdk::PolyFrames FRS, int lsbytes, int iw, int do_dist) { d_split_move_dist(FRS, lsbytes, iw, do_dist); }

// pass 2, detection: the pixel that realises its segment's maximum distance and passes the split tests becomes a
// candidate; everything pass 2 needs from the OLD list is stored with the candidate (the reference reads a snapshot).
// cand record: {i, g, n, maxDist, oldEndIndex, oldRight, x1 bits, y1 bits}
__device__ __forceinline__ void d_split_detect(const rdk::PolyFrames &FRS, int lsbytes, float minerror, int iw, int round) {
  RD_FRAME;
  const ls_rec *ls = (const ls_rec *)FRM.lslist;
  const int *__restrict__ number = s.num[1];
  const int nlive = s.ctr[24];
  SPARSE_LOOP(j, nlive) {
    const int i = s.live[j];
    const int g = s.id[i];
    if (g == 0 || !FITS(g, lsbytes)) continue;
    const ls_rec r = ls[g];
    if (r.polyid == 0) continue;
    if (r.endIndex - r.startIndex < 3) continue;
    if (r.startCount > 1 || r.endCount > 1) continue;
    const int md = r.maxDist;
    if (s.dist[i] != md) continue;
    if (md < (int)(minerror * 65536)) continue;
    if ((float)md < (minerror * 3 * 65536) && (float)md * (float)md / dist2f(r.x0, r.y0, r.x1, r.y1) < 100000.0f) continue;
    const int p = s.pos[i], x = p % iw, y = p / iw;
    if (dist2f((float)x, (float)y, r.x0, r.y0) < 1) continue;
    if (dist2f((float)x, (float)y, r.x1, r.y1) < 1) continue;
    const int c = atomicAdd(&s.ctr[2 + round], 1);
    int *e = s.cand + (size_t)c * 8;
    e[0] = i; e[1] = g; e[2] = number[i]; e[3] = md; e[4] = r.endIndex; e[5] = r.rightPtr;
    e[6] = __float_as_int(r.x1); e[7] = __float_as_int(r.y1);
  }
}
__global__ void k_split_detect(const rdk::PolyFrames FRS, int lsbytes, float minerror, int iw, int round) { d_split_detect(FRS, lsbytes, minerror, iw, round); }

// pass 2, application: new ids follow the raster order of the candidates (rank by pixel index); when several
// candidates share a segment the one latest in raster order decides the shared fields, as a serial execution would.
__device__ __forceinline__ void d_split_apply(const rdk::PolyFrames &FRS, int lsbytes, int iw, int round) {
  RD_FRAME;
  ls_rec *ls = (ls_rec *)FRM.lslist;
  const int C = s.ctr[2 + round];
  const int base = *(const int *)ls;
  SPARSE_LOOP(c, C) {
    const int *e = s.cand + (size_t)c * 8;
    const int i = e[0], g = e[1];
    int rank = 0;
    bool last = true;
    for (int k = 0; k < C; k++) {
      const int *f = s.cand + (size_t)k * 8;
      if (f[0] < i) rank++;
      if (f[1] == g && f[0] > i) last = false;
    }
    const int gn = base + 1 + rank;
    if (!FITS(gn, lsbytes)) continue;
    const int p = s.pos[i], x = p % iw, y = p / iw;
    ls_rec nr = {};
    nr.startIndex = e[2]; nr.endIndex = e[4];
    nr.x0 = (float)x; nr.y0 = (float)y; nr.x1 = __int_as_float(e[6]); nr.y1 = __int_as_float(e[7]);
    nr.leftPtr = g; nr.rightPtr = e[5];
    nr.maxDist = 0; nr.polyid = ls[g].polyid; nr.level = e[3];
    ls[gn] = nr;
    if (last) {
      ls[g].endIndex = e[2]; ls[g].x1 = (float)x; ls[g].y1 = (float)y; ls[g].rightPtr = gn; ls[g].maxDist = 0;
      if (e[5] != 0) ls[e[5]].leftPtr = gn;
    }
  }
}
__global__ void k_split_apply(const rdk::PolyFrames FRS, int lsbytes, int iw, int round) { d_split_apply(FRS, lsbytes, iw, round); }

__device__ __forceinline__ void d_split_commit(const rdk::PolyFrames &FRS, int round) {
  RD_FRAME;
  ls_rec *ls = (ls_rec *)FRM.lslist;
  if (blockIdx.x == 0 && threadIdx.x == 0) *(int *)ls += s.ctr[2 + round];
}
__global__ void k_split_commit(const rdk::PolyFrames FRS, int round) { d_split_commit(FRS, round); }

// ------------------------------------------------------------------------------------------------ refinement (pl:680-809)
__device__ __forceinline__ void d_refine0(const rdk::PolyFrames &FRS, int maxrec) {
  RD_FRAME;
  const ls_rec *ls = (const ls_rec *)FRM.lslist;
  const int n = *(const int *)ls;
  lsx_rec *sx = (lsx_rec *)s.lsx;
  SPARSE_LOOP(g0, n) {
    const int g = g0 + 1;
    if (g >= maxrec) continue;
    lsx_rec z = {};
    if (ls[g].polyid != 0) {
      z.dx = (short)(ls[g].x1 - ls[g].x0); z.dy = (short)(ls[g].y1 - ls[g].y0);
      z.vx = (short)-z.dy; z.vy = z.dx;
      z.d2 = z.dx * z.dx + z.dy * z.dy;
    }
    sx[g] = z;
  }
}
__global__ void k_refine0(const rdk::PolyFrames FRS, int maxrec) { d_refine0(FRS, maxrec); }

__device__ __forceinline__ void d_refine1(const rdk::PolyFrames &FRS, int maxrec, int iw) {
  RD_FRAME;
  const ls_rec *ls = (const ls_rec *)FRM.lslist;
  const int nlive = s.ctr[24];
  const int n = *(const int *)ls;
  lsx_rec *sx = (lsx_rec *)s.lsx;
  SPARSE_LOOP(j, nlive) {
    const int i = s.live[j];
    const int g = s.id[i];
    if (g <= 0 || n < g || g >= maxrec || ls[g].polyid == 0) continue;
    const int p = s.pos[i], x = p % iw, y = p / iw;
    const int vx = x - (int)rintf(ls[g].x0), vy = y - (int)rintf(ls[g].y0);
    const int ay = vx * sx[g].vx + vy * sx[g].vy;
    const int ax0 = vx * sx[g].dx + vy * sx[g].dy;
    const int ax1 = sx[g].d2;
    atomicAdd((unsigned long long *)&sx[g].mx00, (unsigned long long)(long long)rintf((float)ax0 * (float)ax0));
    atomicAdd((unsigned long long *)&sx[g].mx01, (unsigned long long)(long long)rintf((float)ax0 * (float)ax1));
    atomicAdd((unsigned long long *)&sx[g].mx11, (unsigned long long)(long long)rintf((float)ax1 * (float)ax1));
    atomicAdd((unsigned long long *)&sx[g].my0, (unsigned long long)(long long)rintf((float)ax0 * (float)ay));
    atomicAdd((unsigned long long *)&sx[g].my1, (unsigned long long)(long long)rintf((float)ax1 * (float)ay));
  }
}
__global__ void k_refine1(const rdk::PolyFrames FRS, int maxrec, int iw) { d_refine1(FRS, maxrec, iw); }

__device__ __forceinline__ void d_refine2(const rdk::PolyFrames &FRS, int maxrec) {
  RD_FRAME;
  ls_rec *ls = (ls_rec *)FRM.lslist;
  const int n = *(const int *)ls;
  const lsx_rec *sx = (const lsx_rec *)s.lsx;
  SPARSE_LOOP(g0, n) {
    const int g = g0 + 1;
    if (g >= maxrec || ls[g].polyid == 0) continue;
    float rdet = (float)sx[g].mx00 * (float)sx[g].mx11 - (float)sx[g].mx01 * (float)sx[g].mx01;
    if (rdet == 0) continue;
    rdet = (float)(1.0 / (double)rdet);
    const float as0 = ((float)sx[g].mx11 * (float)sx[g].my0 - (float)sx[g].mx01 * (float)sx[g].my1) * rdet;
    const float as1 = ((float)sx[g].mx00 * (float)sx[g].my1 - (float)sx[g].mx01 * (float)sx[g].my0) * rdet;
    ls[g].x0 += (float)sx[g].vx * as1; ls[g].y0 += (float)sx[g].vy * as1;
    ls[g].x1 += (float)sx[g].vx * (as0 + as1); ls[g].y1 += (float)sx[g].vy * (as0 + as1);
  }
}
__global__ void k_refine2(const rdk::PolyFrames FRS, int maxrec) { d_refine2(FRS, maxrec); }

// pl:772-809.  Segment g joins its end point with the start point of its right neighbour h, and h's own step reads that
// start point, so the result depends on the order of the steps; the canonical order is ascending g.
// Steps of non-adjacent segments commute, hence the serial result is obtained by letting a segment run as soon as every
// neighbour with a smaller id has run and keeping neighbours with larger ids waiting: one block, rounds separated by
// barriers; `done` lives in global scratch.
__device__ __forceinline__ void d_refine3(const rdk::PolyFrames &FRS, int maxrec) {
  RD_FRAME;
  ls_rec *ls = (ls_rec *)FRM.lslist;
  __shared__ int progress;
  const int n0 = *(const int *)ls;
  const int n = n0 < maxrec - 1 ? n0 : maxrec - 1;
  int *done = s.dist;   // free at this point; n <= cap
  for (int g = threadIdx.x + 1; g <= n; g += 1024) done[g] = (ls[g].polyid == 0 || ls[g].rightPtr == 0) ? 1 : 0;
  __syncthreads();
  for (int iter = 0; iter <= n; iter++) {
    if (threadIdx.x == 0) progress = 0;
    __syncthreads();
    // decide with the flags as they are at the start of the round
    for (int g = threadIdx.x + 1; g <= n; g += 1024) {
      if (done[g]) continue;
      const int h = ls[g].rightPtr, l = ls[g].leftPtr;
      // every step that touches g's or h's coordinates and has a smaller id must be finished:
      //   left neighbour l (writes g.start), h itself (reads/writes h.start, h.end), h's right neighbour does not touch g or h.start
      const bool l_ok = l <= 0 || l > g || l > n || done[l] == 1;
      const bool h_ok = h > g || h > n || done[h] == 1;
      if (l_ok && h_ok) done[g] = 2;   // runs in this round
    }
    __syncthreads();
    for (int g = threadIdx.x + 1; g <= n; g += 1024) {
      if (done[g] != 2) continue;
      const int h = ls[g].rightPtr;
      const float v0 = ls[g].x0, v1 = ls[g].y0, v2 = ls[g].x1, v3 = ls[g].y1;
      const float u0 = ls[h].x0, u1 = ls[h].y0, u2 = ls[h].x1, u3 = ls[h].y1;
      const float d = (v2 - v0) * (u3 - u1) - (v3 - v1) * (u2 - u0);
      float wx, wy;
      if ((double)fabsf(d) < 1e-6) {
        wx = (v2 + u0) * 0.5f; wy = (v3 + u1) * 0.5f;
      } else {
        const float nn = (v1 - u1) * (u2 - u0) - (v0 - u0) * (u3 - u1);
        const float q = nn / d;
        wx = v0 + q * (v2 - v0); wy = v1 + q * (v3 - v1);
        const float e0 = sqrtf((wx - v2) * (wx - v2) + (wy - v3) * (wy - v3));
        const float e1 = sqrtf((wx - u0) * (wx - u0) + (wy - u1) * (wy - u1));
        if (e0 > 10 && e1 > 10) { wx = (v2 + u0) * 0.5f; wy = (v3 + u1) * 0.5f; }
      }
      ls[g].x1 = wx; ls[g].y1 = wy; ls[h].x0 = wx; ls[h].y0 = wy;
      progress = 1;
    }
    __syncthreads();
    for (int g = threadIdx.x + 1; g <= n; g += 1024) if (done[g] == 2) done[g] = 1;
    __syncthreads();
    if (!progress) break;
    __syncthreads();
  }
}
__global__ __launch_bounds__(1024) void k_refine3(const rdk::PolyFrames FRS, int maxrec) { d_refine3(FRS, maxrec); }

// ------------------------------------------------------------------------------------------------ persistent variant
// Everything from the initial segments to the end-point joining (pl:439-809: about 85 tiny launches above) in ONE
// single-block launch: per-pixel state lives in registers (PP_PX live pixels per thread), segment records and moment
// sums in LDS, phases are separated by block barriers.  Same arithmetic, same tie rules as the kernels above.  If the
// frame does not fit (too many live pixels, records or candidates) the kernel sets ctr[25] and the caller repeats the
// stage with the multi-launch path.
#define PP_T 1024
#define PP_PX 16
#define PP_MAXSEG 1024
#define PP_MAXCAND 448
struct pp_lds {
  ls_rec rec[PP_MAXSEG];
  lsx_rec sx[PP_MAXSEG];
  int start_i[PP_MAXSEG], end_i[PP_MAXSEG];
  int done[PP_MAXSEG];
  int cand[PP_MAXCAND * 8];
  int ncand, count, fail, progress;
};

__device__ __noinline__ int2 pp_move_dist(pp_lds &L, int pk, int xyv, int dreg, bool move_only, int rnd13, int round) {
  int g = pk & 0x7ff;
  // a segment untouched by the previous round keeps its end index, its maximum and its pixels' distances: nothing to redo
  if (L.done[g] != round) return make_int2(pk, dreg);
  const int num = (int)((unsigned)pk >> 11);
  if (g == 0 || L.rec[g].polyid == 0) return make_int2(pk, dreg);
  if (L.rec[g].endIndex < num) { g = L.rec[g].rightPtr; pk = (pk & ~0x7ff) | g; }
  if (move_only) return make_int2(pk, dreg);
  if (g == 0 || L.rec[g].polyid == 0) return make_int2(pk, dreg);
  const int x = xyv & 0xffff, y = xyv >> 16;
  const float vx = (float)(int)L.rec[g].x0, vy = (float)(int)L.rec[g].y0, wx = (float)(int)L.rec[g].x1, wy = (float)(int)L.rec[g].y1;
  float cx, cy;
  const float l2 = dist2f(vx, vy, wx, wy);
  if (l2 <= 1e-4f) { cx = vx; cy = vy; }
  else {
    const float t = (((float)x - vx) * (wx - vx) + ((float)y - vy) * (wy - vy)) / l2;
    if (t < 0.0f) { cx = vx; cy = vy; }
    else if (t > 1.0f) { cx = wx; cy = wy; }
    else { cx = vx + t * (wx - vx); cy = vy + t * (wy - vy); }
  }
  const float a = cx - (float)x, b = cy - (float)y;
  int d = (int)((float)sqrt((double)a * (double)a + (double)b * (double)b) * 65536);
  d ^= rnd13;                  // pixel_rand(pixel, 0) & 0x1fff: the same in every round, computed once (see the kernel)
  dreg = d;
  atomicMax(&L.rec[g].maxDist, d);
  return make_int2(pk, dreg);
}

// (reads the list as it is: nothing is modified in this phase)
__device__ __noinline__ void pp_detect(pp_lds &L, int pk, int xyv, int dreg, float minerror, const int *live, int j) {
  const int g = pk & 0x7ff;
  const ls_rec r = L.rec[g];
  if (r.polyid == 0) return;
  if (r.endIndex - r.startIndex < 3) return;
  if (r.startCount > 1 || r.endCount > 1) return;
  const int md = r.maxDist;
  if (dreg != md) return;
  if (md < (int)(minerror * 65536)) return;
  if ((float)md < (minerror * 3 * 65536) && (float)md * (float)md / dist2f(r.x0, r.y0, r.x1, r.y1) < 100000.0f) return;
  const int x = xyv & 0xffff, y = xyv >> 16;
  if (dist2f((float)x, (float)y, r.x0, r.y0) < 1) return;
  if (dist2f((float)x, (float)y, r.x1, r.y1) < 1) return;
  const int c = atomicAdd(&L.ncand, 1);
  if (c >= PP_MAXCAND) { L.fail = 1; return; }
  int *e = L.cand + c * 8;
  e[0] = live[j]; e[1] = g; e[2] = (int)((unsigned)pk >> 11); e[3] = md; e[4] = r.endIndex; e[5] = r.rightPtr;
  e[6] = __float_as_int(r.x1); e[7] = __float_as_int(r.y1);
}

__device__ __noinline__ void pp_moments(pp_lds &L, int pk, int xyv, int n) {
  const int g = pk & 0x7ff;
  if (g <= 0 || n < g || L.rec[g].polyid == 0) return;
  const int vx = (xyv & 0xffff) - (int)rintf(L.rec[g].x0), vy = (xyv >> 16) - (int)rintf(L.rec[g].y0);
  const int ay = vx * L.sx[g].vx + vy * L.sx[g].vy;
  const int ax0 = vx * L.sx[g].dx + vy * L.sx[g].dy;
  const int ax1 = L.sx[g].d2;
  atomicAdd((unsigned long long *)&L.sx[g].mx00, (unsigned long long)(long long)rintf((float)ax0 * (float)ax0));
  atomicAdd((unsigned long long *)&L.sx[g].mx01, (unsigned long long)(long long)rintf((float)ax0 * (float)ax1));
  atomicAdd((unsigned long long *)&L.sx[g].mx11, (unsigned long long)(long long)rintf((float)ax1 * (float)ax1));
  atomicAdd((unsigned long long *)&L.sx[g].my0, (unsigned long long)(long long)rintf((float)ax0 * (float)ay));
  atomicAdd((unsigned long long *)&L.sx[g].my1, (unsigned long long)(long long)rintf((float)ax1 * (float)ay));
}

__global__ __launch_bounds__(PP_T) void k_poly_persistent(const rdk::PolyFrames FRS, int lsbytes, float minerror, int iw) {
  RD_FRAME;
  ls_rec *ls = (ls_rec *)FRM.lslist;
  const int *__restrict__ number = s.num[1];
  extern __shared__ __attribute__((aligned(16))) char pp_raw[];
  pp_lds &L = *(pp_lds *)pp_raw;
  const int tid = threadIdx.x;
  const int nlive = s.ctr[24], K = s.ctr[1];
  const int maxrec = lsbytes / 56;
  if (tid == 0) s.ctr[39] = (int)wall_clock64();   // ctr[39..45]: 100 MHz time stamps of the phases (diagnostics)
  if (nlive > PP_T * PP_PX || K >= PP_MAXSEG - 1 || K >= maxrec - 1) { if (tid == 0) s.ctr[25] = 1; return; }

  // per-pixel state in registers: id (11 bits) | position along the chain << 11, x | y << 16, last distance
  int pk[PP_PX], xy[PP_PX], dreg[PP_PX], rnd[PP_PX / 2];
#define PP_ID(k) (pk[k] & 0x7ff)
#define PP_NUM(k) ((int)((unsigned)pk[k] >> 11))
#define PP_CI(k) (s.live[tid + (k) * PP_T])
#define PP_SEQ __builtin_amdgcn_sched_barrier(0)
  {   // two levels of gathers, each issued for all PP_PX pixels together
    int cc[PP_PX];
#pragma unroll
    for (int k = 0; k < PP_PX; k++) { const int j = tid + k * PP_T; cc[k] = s.live[j < nlive ? j : 0]; }
#pragma unroll
    for (int k = 0; k < PP_PX; k++) { const int j = tid + k * PP_T; if (j >= nlive) cc[k] = 0; }
#pragma unroll
    for (int k = 0; k < PP_PX; k++) { xy[k] = s.pos[cc[k]]; pk[k] = s.id[cc[k]]; dreg[k] = number[cc[k]]; }
#pragma unroll
    for (int k = 0; k < PP_PX; k++) {
      const int j = tid + k * PP_T;
      const int p = xy[k];
      pk[k] = j < nlive ? (pk[k] | (dreg[k] << 11)) : 0;
      xy[k] = j < nlive ? ((p % iw) | ((p / iw) << 16)) : 0;
      dreg[k] = 0;
      // the tie-breaking bits of the pixel's distance (pl:870-889 mixing function): two per register
      if ((k & 1) == 0) rnd[k >> 1] = 0;
      rnd[k >> 1] |= (j < nlive ? (pixel_rand(p, 0) & 0x1fff) : 0) << (16 * (k & 1));
      PP_SEQ;
    }
  }
  for (int g = tid; g < PP_MAXSEG; g += PP_T) L.done[g] = 0;   // during the rounds: round in which the segment has to be looked at again
  for (int g = tid; g <= K; g += PP_T) { ls_rec z = {}; L.rec[g] = z; L.start_i[g] = -1; L.end_i[g] = 0x7fffffff; }
  if (tid == 0) { L.ncand = 0; L.count = K; L.fail = 0; L.progress = 0; }
  __syncthreads();

  if (tid == 0) s.ctr[40] = (int)wall_clock64();
  // initial segments (pl:439-506)
#pragma unroll
  for (int k = 0; k < PP_PX; k++) {
    const int g = PP_ID(k);
    if (g != 0) {
      if (PP_NUM(k) == 1) { atomicAdd(&L.rec[g].startCount, 1); atomicMax(&L.start_i[g], PP_CI(k)); }
      atomicAdd(&L.rec[g].npix, 1);
      atomicMax(&L.rec[g].endIndex, PP_NUM(k));
    }
    PP_SEQ;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PP_PX; k++) {
    const int g = PP_ID(k);
    if (g != 0 && PP_NUM(k) == L.rec[g].endIndex && L.rec[g].startCount == 1 && L.rec[g].npix >= 2) { atomicAdd(&L.rec[g].endCount, 1); atomicMin(&L.end_i[g], PP_CI(k)); }
    PP_SEQ;
  }
  __syncthreads();
  for (int g = tid + 1; g <= K; g += PP_T) {
    const int sp = L.start_i[g], ep = L.end_i[g];
    if (sp >= 0) { const int p = s.pos[sp]; L.rec[g].x0 = (float)(p % iw); L.rec[g].y0 = (float)(p / iw); }
    if (L.rec[g].startCount == 1 && L.rec[g].npix >= 2 && ep != 0x7fffffff) {
      const int p = s.pos[ep];
      L.rec[g].x1 = (float)(p % iw); L.rec[g].y1 = (float)(p / iw);
      L.rec[g].polyid = g;
    } else L.rec[g].polyid = 0;
  }
  __syncthreads();

  if (tid == 0) s.ctr[41] = (int)wall_clock64();
  // subdivision rounds (pl:509-646); round 15 only moves pixels
  for (int round = 0; round <= 15; round++) {
    if (tid == 0) { s.ctr[46 + round] = (int)wall_clock64(); s.ctr[38] = round + 1; }      // (diagnostics: start of every round, rounds run)
    // (pixels in segments that the previous round did not change return at once: see pp_move_dist)
#pragma unroll
    for (int k = 0; k < PP_PX; k++) {
      const int2 r2 = pp_move_dist(L, pk[k], xy[k], dreg[k], round == 15, (rnd[k >> 1] >> (16 * (k & 1))) & 0x1fff, round);
      pk[k] = r2.x; dreg[k] = r2.y;
      PP_SEQ;
    }
    __syncthreads();
    if (round == 15) break;
    // detection (reads the list as it is: nothing is modified in this phase)
#pragma unroll
    for (int k = 0; k < PP_PX; k++) {
      if (PP_ID(k) != 0 && L.done[PP_ID(k)] == round) pp_detect(L, pk[k], xy[k], dreg[k], minerror, s.live, tid + k * PP_T);   // others: same verdict as last round (none)
      PP_SEQ;
    }
    __syncthreads();
    if (L.ncand == 0) break;     // no split this round: the remaining rounds would find the same (block-uniform)
    __syncthreads();
    {
      const int C = L.ncand < PP_MAXCAND ? L.ncand : PP_MAXCAND;
      const int base = L.count;
      for (int c = tid; c < C; c += PP_T) {
        const int *e = L.cand + c * 8;
        const int i = e[0], g = e[1];
        int rank = 0;
        bool last = true;
        for (int q = 0; q < C; q++) {
          const int *f = L.cand + q * 8;
          if (f[0] < i) rank++;
          if (f[1] == g && f[0] > i) last = false;
        }
        const int gn = base + 1 + rank;
        if (gn >= PP_MAXSEG || gn >= maxrec - 1) { L.fail = 1; continue; }
        const int p = s.pos[i], x = p % iw, y = p / iw;
        ls_rec nr = {};
        nr.startIndex = e[2]; nr.endIndex = e[4];
        nr.x0 = (float)x; nr.y0 = (float)y; nr.x1 = __int_as_float(e[6]); nr.y1 = __int_as_float(e[7]);
        nr.leftPtr = g; nr.rightPtr = e[5];
        nr.maxDist = 0; nr.polyid = L.rec[g].polyid; nr.level = e[3];
        L.rec[gn] = nr;
        L.done[gn] = round + 1;
        if (last) {
          L.done[g] = round + 1;
          L.rec[g].endIndex = e[2]; L.rec[g].x1 = (float)x; L.rec[g].y1 = (float)y; L.rec[g].rightPtr = gn; L.rec[g].maxDist = 0;
          if (e[5] != 0) L.rec[e[5]].leftPtr = gn;
        }
      }
    }
    __syncthreads();
    if (tid == 0) { const int C = L.ncand < PP_MAXCAND ? L.ncand : PP_MAXCAND; L.count = (L.count + C < PP_MAXSEG - 1) ? L.count + C : PP_MAXSEG - 1; L.ncand = 0; }
    __syncthreads();
  }

  if (tid == 0) s.ctr[42] = (int)wall_clock64();
  // refinement (pl:680-809)
  const int n = L.count;
  for (int g = tid + 1; g <= n; g += PP_T) {
    lsx_rec z = {};
    if (L.rec[g].polyid != 0) {
      z.dx = (short)(L.rec[g].x1 - L.rec[g].x0); z.dy = (short)(L.rec[g].y1 - L.rec[g].y0);
      z.vx = (short)-z.dy; z.vy = z.dx;
      z.d2 = z.dx * z.dx + z.dy * z.dy;
    }
    L.sx[g] = z;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PP_PX; k++) {
    pp_moments(L, pk[k], xy[k], n);
    PP_SEQ;
  }
  __syncthreads();
  for (int g = tid + 1; g <= n; g += PP_T) {
    if (L.rec[g].polyid == 0) continue;
    const lsx_rec q = L.sx[g];
    float rdet = (float)q.mx00 * (float)q.mx11 - (float)q.mx01 * (float)q.mx01;
    if (rdet == 0) continue;
    rdet = (float)(1.0 / (double)rdet);
    const float as0 = ((float)q.mx11 * (float)q.my0 - (float)q.mx01 * (float)q.my1) * rdet;
    const float as1 = ((float)q.mx00 * (float)q.my1 - (float)q.mx01 * (float)q.my0) * rdet;
    L.rec[g].x0 += (float)q.vx * as1; L.rec[g].y0 += (float)q.vy * as1;
    L.rec[g].x1 += (float)q.vx * (as0 + as1); L.rec[g].y1 += (float)q.vy * (as0 + as1);
  }
  __syncthreads();
  if (tid == 0) s.ctr[43] = (int)wall_clock64();
  // end-point joining in dependency order (see k_refine3)
  for (int g = tid + 1; g <= n; g += PP_T) L.done[g] = (L.rec[g].polyid == 0 || L.rec[g].rightPtr == 0) ? 1 : 0;
  __syncthreads();
  for (int iter = 0; iter <= n; iter++) {
    if (tid == 0) L.progress = 0;
    __syncthreads();
    for (int g = tid + 1; g <= n; g += PP_T) {
      if (L.done[g]) continue;
      const int h = L.rec[g].rightPtr, l = L.rec[g].leftPtr;
      const bool l_ok = l <= 0 || l > g || l > n || L.done[l] == 1;
      const bool h_ok = h > g || h > n || L.done[h] == 1;
      if (l_ok && h_ok) L.done[g] = 2;
    }
    __syncthreads();
    for (int g = tid + 1; g <= n; g += PP_T) {
      if (L.done[g] != 2) continue;
      const int h = L.rec[g].rightPtr;
      const float v0 = L.rec[g].x0, v1 = L.rec[g].y0, v2 = L.rec[g].x1, v3 = L.rec[g].y1;
      const float u0 = L.rec[h].x0, u1 = L.rec[h].y0, u2 = L.rec[h].x1, u3 = L.rec[h].y1;
      const float d = (v2 - v0) * (u3 - u1) - (v3 - v1) * (u2 - u0);
      float wx, wy;
      if ((double)fabsf(d) < 1e-6) {
        wx = (v2 + u0) * 0.5f; wy = (v3 + u1) * 0.5f;
      } else {
        const float nn = (v1 - u1) * (u2 - u0) - (v0 - u0) * (u3 - u1);
        const float q = nn / d;
        wx = v0 + q * (v2 - v0); wy = v1 + q * (v3 - v1);
        const float e0 = sqrtf((wx - v2) * (wx - v2) + (wy - v3) * (wy - v3));
        const float e1 = sqrtf((wx - u0) * (wx - u0) + (wy - u1) * (wy - u1));
        if (e0 > 10 && e1 > 10) { wx = (v2 + u0) * 0.5f; wy = (v3 + u1) * 0.5f; }
      }
      L.rec[g].x1 = wx; L.rec[g].y1 = wy; L.rec[h].x0 = wx; L.rec[h].y0 = wy;
      L.progress = 1;
    }
    __syncthreads();
    for (int g = tid + 1; g <= n; g += PP_T) if (L.done[g] == 2) L.done[g] = 1;
    __syncthreads();
    if (!L.progress) break;
    __syncthreads();
  }

  if (tid == 0) s.ctr[44] = (int)wall_clock64();
  // results: record list (header = count) and the final per-pixel ids
  if (tid == 0) { ls_rec z = {}; ls[0] = z; *(int *)ls = n; if (L.fail) s.ctr[25] = 1; }
  for (int g = tid + 1; g <= n; g += PP_T) ls[g] = L.rec[g];
#pragma unroll
  for (int k = 0; k < PP_PX; k++) if (tid + k * PP_T < nlive) s.id[PP_CI(k)] = PP_ID(k);
  if (tid == 0) s.ctr[45] = (int)wall_clock64();
}

__global__ void k_scatter_ids(const rdk::PolyFrames FRS) {
  RD_FRAME;
  int *ids = FRM.ids;
  const int nlive = s.ctr[24];
  SPARSE_LOOP(j, nlive) { const int i = s.live[j]; ids[s.pos[i]] = s.id[i]; }
}

template <typename T> T *dalloc(size_t n) { void *p = nullptr; if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) return nullptr; return (T *)p; }

}  // namespace

namespace rdk {

PolyScratch *poly_scratch_create(int iw, int ih) {
  PolyScratch *ps = new PolyScratch();
  const size_t N = (size_t)iw * ih;
  ps->cap = (int)N;
  const size_t NWORDS = (size_t)cdiv(iw, 64) * ih + 8;
  ps->sb = dalloc<unsigned long long>(NWORDS); ps->rb = dalloc<unsigned long long>(NWORDS); ps->tb = dalloc<unsigned long long>(NWORDS);
  ps->pw = dalloc<int>(NWORDS); ps->rowsum = dalloc<int>(ih + 8); ps->rowbase = dalloc<int>(ih + 8);
  ps->cstate = dalloc<unsigned long long>(3 * (N / CP_PER_BLOCK + 2)); ps->csync = dalloc<int>(4);
  ps->pos = dalloc<int>(N); ps->nbr = dalloc<int>(N * 8);
  ps->lab = dalloc<int>(N); ps->alive = dalloc<int>(N); ps->ends = dalloc<int>(N);
  for (int k = 0; k < 2; k++) { ps->nx[k] = dalloc<int>(N); ps->pv[k] = dalloc<int>(N); ps->num[k] = dalloc<int>(N); ps->link[k] = dalloc<int>(N); }
  ps->flag = dalloc<int>(N); ps->flag2 = dalloc<int>(N);
  ps->lab2 = dalloc<int>(N); ps->size = dalloc<int>(N); ps->rootid = dalloc<int>(N); ps->id = dalloc<int>(N); ps->dist = dalloc<int>(N + 2);
  ps->cand = dalloc<int>(N * 8 / 4 + 64);
  ps->ctr = dalloc<int>(64);
  ps->lsx = dalloc<lsx_rec>(N * 16 / 56 + 2);
  ps->segaux = dalloc<int>(2 * (N * 16 / 56 + 2));
  ps->live = dalloc<int>(N);
  (void)hipMemset(ps->ctr, 0, 64 * sizeof(int));
  (void)hipMemset(ps->cstate, 0, 3 * (N / CP_PER_BLOCK + 2) * sizeof(unsigned long long));
  (void)hipMemset(ps->csync, 0, 4 * sizeof(int));      // [0]: generation of the compaction state words (k_tidy_bits advances it before use)
  (void)hipStreamSynchronize(0);   // the fill is asynchronous and the callers' streams do not wait for the null stream
  return ps;
}

const int *poly_scratch_counters(const PolyScratch *ps) { return ps->ctr; }

void poly_scratch_destroy(PolyScratch *ps) {
  if (!ps) return;
  void *all[] = { ps->sb, ps->rb, ps->tb, ps->pw, ps->rowsum, ps->rowbase, ps->cstate, ps->csync, ps->pos, ps->nbr, ps->lab, ps->alive, ps->ends, ps->nx[0], ps->nx[1], ps->pv[0], ps->pv[1],
                  ps->num[0], ps->num[1], ps->link[0], ps->link[1], ps->flag, ps->flag2, ps->lab2, ps->size, ps->rootid, ps->id, ps->dist, ps->cand, ps->ctr, ps->lsx, ps->segaux, ps->live };
  for (void *p : all) if (p) (void)hipFree(p);
  delete ps;
}

// frames_host: nb <= RD_MAXB descriptors (frame z of every launch = frames_host[z]); all frames share the scalar parameters
void polyline(hipStream_t st, const PolyFrame *frames_host, int nb, int lslist_bytes, int ring_const, float minerror, int sizeThre, int iw, int ih, int mode) {
  const PolyFrames frames = pack_frames(frames_host, nb);
  const int N = iw * ih;
  const int maxrec = lslist_bytes / 56;
  const dim3 sg(SPARSE_GRID, 1, nb), sb(256);

  // tidy (oclpolyline.c:222-235) and compaction of the chain pixels in raster order, on bit planes
  const int wpr = cdiv(iw, 64);
  // (all frames of a launch come from one caller: either all hand over the strong mask as a bit plane - then the stale-ring emulation of H3 takes its constant form and a
  //  ring source cannot be honoured - or all as int planes)
  for (int z = 0; z < nb; z++) {
    if ((frames_host[z].in_bits == nullptr) != (frames_host[0].in_bits == nullptr)) { fprintf(stderr, "polyline: the frames of one launch mix bit-plane and int-plane input\n"); abort(); }
    if (frames_host[z].in_bits != nullptr && frames_host[z].ring_src != nullptr) { fprintf(stderr, "polyline: a frame hands over a bit plane AND a ring source (the ring is only read from int planes)\n"); abort(); }
  }
  if (frames_host[0].in_bits == nullptr) hipLaunchKernelGGL(k_mask_bits, dim3(wpr, cdiv(ih, 4), nb), dim3(64, 4), 0, st, frames, iw, ih, wpr);      // (all frames of a launch come from one caller)
  hipLaunchKernelGGL(k_tidy_bits, dim3(cdiv(wpr, 4), cdiv(ih, TB_OUT), nb), dim3(256), 0, st, frames, ring_const, iw, ih, wpr);
  hipLaunchKernelGGL(k_row_prefix, dim3(cdiv(ih, 4), 1, nb), dim3(256), 0, st, frames, ih, wpr);
  hipLaunchKernelGGL(k_chain_scatter, dim3(cdiv(ih, CS_ROWS_PER_BLOCK), 1, nb), dim3(256), 0, st, frames, iw, ih, wpr);
  const int nblk = cdiv(N, CP_PER_BLOCK);
  const dim3 cg(nblk, 1, nb);

  // chains, loops, ends (oclpolyline.c:237-266)
  hipLaunchKernelGGL(k_chain_union, sg, sb, 0, st, frames, iw, wpr);
  hipLaunchKernelGGL(k_flatten, sg, sb, 0, st, frames, 0);      // + ends per chain
  hipLaunchKernelGGL(k_find_ends0, sg, sb, 0, st, frames);
  hipLaunchKernelGGL(k_find_ends0_flags, sg, sb, 0, st, frames);
  for (int r = 0; r < 4; r++) hipLaunchKernelGGL(k_find_ends1, sg, sb, 0, st, frames, r & 1, r == 3 ? 1 : 0);      // (the last one also links and numbers: pl:269-285)
  // numbering (oclpolyline.c:268-275): three rounds 0->1->0->1 (the result is num[1])
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_number, sg, sb, 0, st, frames, r & 1, r == 2 ? 1 : 0);

  // split at numbering jumps, size filter, compact ids (oclpolyline.c:277-295)
  hipLaunchKernelGGL(k_sub_union, sg, sb, 0, st, frames);
  hipLaunchKernelGGL(k_flatten, sg, sb, 0, st, frames, 1);     // + sub-chain sizes
  // (the chain-pixel count lives on the device: launch for the worst case, blocks beyond it exit at once)
  hipLaunchKernelGGL(k_compact1<1>, cg, dim3(256), 0, st, frames, N, nblk, sizeThre, 0);
  hipLaunchKernelGGL(k_compact1<2>, cg, dim3(256), 0, st, frames, N, nblk, 0, mode == 1 ? 1 : 0);

  if (mode == 1) {
    // fast path: initial segments, 15 subdivision rounds and the refinement in one persistent launch, one block per frame (overflow -> ctr[25])
    static std::atomic<unsigned> lds_set{0};
    set_max_lds_once((const void *)k_poly_persistent, (int)sizeof(pp_lds), lds_set);
    hipLaunchKernelGGL(k_poly_persistent, dim3(1, 1, nb), dim3(PP_T), sizeof(pp_lds), st, frames, lslist_bytes, minerror, iw);
    return;
  }

  // initial segments (oclpolyline.c:191-197)
  hipLaunchKernelGGL(k_seg_clear, sg, sb, 0, st, frames, lslist_bytes);
  hipLaunchKernelGGL(k_seg_pass0a, sg, sb, 0, st, frames, lslist_bytes);
  hipLaunchKernelGGL(k_seg_pass0b, sg, sb, 0, st, frames, lslist_bytes);
  hipLaunchKernelGGL(k_seg_finish, sg, sb, 0, st, frames, lslist_bytes, iw);

  // 15 subdivision rounds (oclpolyline.c:202-213)
  for (int r = 0; r < 15; r++) {
    hipLaunchKernelGGL(k_split_move_dist, sg, sb, 0, st, frames, lslist_bytes, iw, 1);
    hipLaunchKernelGGL(k_split_detect, sg, sb, 0, st, frames, lslist_bytes, minerror, iw, r);
    hipLaunchKernelGGL(k_split_apply, sg, sb, 0, st, frames, lslist_bytes, iw, r);
    hipLaunchKernelGGL(k_split_commit, dim3(1, 1, nb), dim3(64), 0, st, frames, r);
  }
  hipLaunchKernelGGL(k_split_move_dist, sg, sb, 0, st, frames, lslist_bytes, iw, 0);

  // refinement (oclpolyline.c:299-306)
  hipLaunchKernelGGL(k_refine0, sg, sb, 0, st, frames, maxrec);
  hipLaunchKernelGGL(k_refine1, sg, sb, 0, st, frames, maxrec, iw);
  hipLaunchKernelGGL(k_refine2, sg, sb, 0, st, frames, maxrec);
  hipLaunchKernelGGL(k_refine3, dim3(1, 1, nb), dim3(1024), 0, st, frames, maxrec);
}

// per-pixel segment ids as a dense plane (lsIdOut = frames[z].ids) from the compact state
void polyline_ids(hipStream_t st, const PolyFrame *frames_host, int nb, int n) {
  const PolyFrames frames = pack_frames(frames_host, nb);
  for (int z = 0; z < nb; z++) (void)hipMemsetAsync(frames_host[z].ids, 0, sizeof(int) * (size_t)n, st);
  hipLaunchKernelGGL(k_scatter_ids, dim3(SPARSE_GRID, 1, nb), dim3(256), 0, st, frames);
}

}  // namespace rdk
