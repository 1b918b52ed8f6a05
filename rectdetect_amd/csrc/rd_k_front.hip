// rectdetect-mi355x: front-end kernels for gfx950 - colour conversion, sigma=1 IIR Gaussian, gradient
// direction, edge strength, bicubic non-max suppression and the element-wise operators.
//
// Reference behaviour being reproduced (file:line into the reference): oclimgutil.cl ("iu").
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off  (see rd_device.h for the arithmetic contract).
#include <stdlib.h>
#include "rd_device.h"
#include "rd_kernels.h"

#define RD_LUT_ATTR __device__
#include "rd_luts.h"

namespace {

using namespace rd;

struct P3 { float *p[3]; };
struct P3c { const float *p[3]; };
struct SrcZ { const uint8_t *p[RD_ZB_MAX]; };      // the source frames of a group launch (caller's buffers: no common pitch)

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline dim3 grid2(int iw, int ih) { return dim3(cdiv(iw, 64), cdiv(ih, 4)); }
const dim3 block2(64, 4);
inline int grid1(int n) { int g = cdiv(n, 256); return g < 1 ? 1 : g; }

// ------------------------------------------------------------------------------------------------ colour
// iu:106-134.  Integer-only sRGB -> packed Lab.  Matrix entries are (int)(m*16384+0.5f) of the sRGB->XYZ matrix,
// 34476 / 30097 = (int)(32768/xn + 0.5f), (int)(32768/zn + 0.5f); the three LUTs are staged in LDS because every
// lane indexes them with a different colour.
__device__ __forceinline__ int lerp_lut(const unsigned short *t, int c) { return t[c >> 8] * (256 - (c & 255)) + t[(c >> 8) + 1] * (c & 255); }

__global__ __launch_bounds__(256) void k_bgr2plab(uint32_t *__restrict__ out, const uint8_t *__restrict__ bgr, int iw, int ih, int ws) {
  __shared__ unsigned short s_s2l[RD_LUT_S2L_N], s_cf[RD_LUT_CF_N], s_cf2[RD_LUT_CF_N];
  const int tid = rd_ty() * 64 + threadIdx.x;
  for (int i = tid; i < RD_LUT_S2L_N; i += 256) s_s2l[i] = rd_lut_s2l[i];
  for (int i = tid; i < RD_LUT_CF_N; i += 256) { s_cf[i] = rd_lut_cfunc[i]; s_cf2[i] = rd_lut_cfunc2[i]; }
  __syncthreads();
  const int x = blockIdx.x * 64 + threadIdx.x;
  if (x >= iw) return;
  // 4 rows per thread-row so that the LUT staging is amortised over 1024 pixels per block
  for (int r = 0; r < 4; r++) {
    const int y = (blockIdx.y * 4 + rd_ty()) * 4 + r;
    if (y >= ih) break;
    const uint8_t *p = bgr + (size_t)y * ws + x * 3;
    const int ib = s_s2l[p[0]], ig = s_s2l[p[1]], ir = s_s2l[p[2]];
    const int cx = (((ir * 6758 + ig * 5859 + ib * 2956 + (1 << 14)) >> 15) * 34476 + (1 << 10)) >> 11;
    const int cy = ((ir * 3484 + ig * 11717 + ib * 1182) + (1 << 10)) >> 11;
    const int cz = (((ir * 317 + ig * 1953 + ib * 15569 + (1 << 14)) >> 15) * 30097 + (1 << 10)) >> 11;
    const int cl = ((lerp_lut(s_cf2, cy) >> 12) + 1) >> 1;
    const int fx = lerp_lut(s_cf, cx), fy = lerp_lut(s_cf, cy), fz = lerp_lut(s_cf, cz);
    const int fxy = (fx - fy + (1 << 7)) >> 8, fyz = (fy - fz + (1 << 7)) >> 8;
    const int ca = (fxy * 8031 + (134744072 + (1 << 17))) >> 18;
    const int cb = (fyz * 3213 + (134744072 + (1 << 17))) >> 18;
    uint32_t v = clampu((uint32_t)cb, 0u, 1023u);
    v = (v << 10) | clampu((uint32_t)ca, 0u, 1023u);
    v = (v << 12) | clampu((uint32_t)cl, 0u, 4095u);
    out[y * iw + x] = v;
  }
}

// the same conversion for one 64x64 tile, which also leaves the L, a, b planes TRANSPOSED (input of the first blur sweep) -
// saves re-reading the packed plane and one launch per frame.  The transposed planes hold the integer FIELDS (16 bits each; the
// sweep turns them into the floats of iu:36-39 as it loads them, iir_field): the blocked sweep reads its source three times over,
// and those reads come from HBM.
__global__ __launch_bounds__(256) void k_bgr2plab_t(uint32_t *__restrict__ out, P3 dst, SrcZ srcz, int iw, int ih, int ws, size_t zs) {
  const uint8_t *__restrict__ bgr = srcz.p[blockIdx.z];
  RD_ZSHIFT(zs, out, dst.p[0], dst.p[1], dst.p[2]);
  __shared__ unsigned short s_s2l[RD_LUT_S2L_N], s_cf[RD_LUT_CF_N], s_cf2[RD_LUT_CF_N];
  __shared__ unsigned short tile[3][64][66];
  const int tid = rd_ty() * 64 + threadIdx.x;
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 64;
  const int x = x0 + threadIdx.x;
  // the thread's 16 pixels: all of their bytes are requested before the first is used (one wait for memory instead of sixteen) - and before the tables are
  // staged, so that the frame's bytes travel while the tables do (the tables' loops wait for memory three times over)
  uint8_t pb[16], pg[16], pr[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int y = y0 + rd_ty() + 4 * k;
    const uint8_t *p = bgr + ((x < iw && y < ih) ? (size_t)y * ws + x * 3 : 0);
    pb[k] = p[0]; pg[k] = p[1]; pr[k] = p[2];
  }
  for (int i = tid; i < RD_LUT_S2L_N; i += 256) s_s2l[i] = rd_lut_s2l[i];
  for (int i = tid; i < RD_LUT_CF_N; i += 256) { s_cf[i] = rd_lut_cfunc[i]; s_cf2[i] = rd_lut_cfunc2[i]; }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int r = rd_ty() + 4 * k;
    const int y = y0 + r;
    if (x >= iw || y >= ih) continue;
    const int ib = s_s2l[pb[k]], ig = s_s2l[pg[k]], ir = s_s2l[pr[k]];
    const int cx = (((ir * 6758 + ig * 5859 + ib * 2956 + (1 << 14)) >> 15) * 34476 + (1 << 10)) >> 11;
    const int cy = ((ir * 3484 + ig * 11717 + ib * 1182) + (1 << 10)) >> 11;
    const int cz = (((ir * 317 + ig * 1953 + ib * 15569 + (1 << 14)) >> 15) * 30097 + (1 << 10)) >> 11;
    const int cl = ((lerp_lut(s_cf2, cy) >> 12) + 1) >> 1;
    const int fx = lerp_lut(s_cf, cx), fy = lerp_lut(s_cf, cy), fz = lerp_lut(s_cf, cz);
    const int fxy = (fx - fy + (1 << 7)) >> 8, fyz = (fy - fz + (1 << 7)) >> 8;
    const int ca = (fxy * 8031 + (134744072 + (1 << 17))) >> 18;
    const int cb = (fyz * 3213 + (134744072 + (1 << 17))) >> 18;
    uint32_t v = clampu((uint32_t)cb, 0u, 1023u);
    v = (v << 10) | clampu((uint32_t)ca, 0u, 1023u);
    v = (v << 12) | clampu((uint32_t)cl, 0u, 4095u);
    out[y * iw + x] = v;
    tile[0][r][threadIdx.x] = (unsigned short)(v & 4095u); tile[1][r][threadIdx.x] = (unsigned short)((v >> 12) & 1023u); tile[2][r][threadIdx.x] = (unsigned short)((v >> 22) & 1023u);
  }
  __syncthreads();
  for (int r = rd_ty(); r < 64; r += 4) {
    const int ox = y0 + threadIdx.x, oy = x0 + r;   // output planes are ih wide, iw tall
    if (ox < ih && oy < iw)
      for (int k = 0; k < 3; k++) ((unsigned short *)dst.p[k])[(size_t)oy * ih + ox] = tile[k][threadIdx.x][r];
  }
}

// four independent elements per thread and iteration: these kernels are pure memory streams, and one 4-byte load in
// flight per lane leaves most of the HBM pipeline idle
template <typename LD, typename ST> __device__ __forceinline__ void ew4(int n, LD ld, ST st) {
  const int stride = gridDim.x * blockDim.x;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const auto v0 = ld(i), v1 = ld(i + stride), v2 = ld(i + 2 * stride), v3 = ld(i + 3 * stride);
    st(i, v0); st(i + stride, v1); st(i + 2 * stride, v2); st(i + 3 * stride, v3);
  }
  for (; i < n; i += stride) st(i, ld(i));
}

// iu:333-342 / iu:325-331
__global__ void k_unpack_plab(float *__restrict__ L, float *__restrict__ a, float *__restrict__ b, const uint32_t *__restrict__ in, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float l, aa, bb;
    unpack_lab(in[i], l, aa, bb);
    L[i] = l; a[i] = aa; b[i] = bb;
  }
}

__global__ void k_pack_plab(uint32_t *__restrict__ out, const float *__restrict__ L, const float *__restrict__ a, const float *__restrict__ b, int n) {
  ew4(n, [=](int i) { return make_float3(L[i], a[i], b[i]); }, [=](int i, float3 v) { out[i] = pack_lab(v.x, v.y, v.z); });
}

// ------------------------------------------------------------------------------------------------ transposes
// 64x64 tiles through LDS (row pitch 65 floats -> conflict-free column reads); block = 64x4 threads.
template <int MODE>  // 0: plain float planes, 1: source is packed Lab (unpack while transposing), 2: IIR combine while transposing
__global__ __launch_bounds__(256) void k_transpose(P3 dst, P3c src, P3c fwd, P3c bwd, const uint32_t *__restrict__ plab, int np, int W, int H, const int *only_if) {
  if (only_if && *only_if == 0) return;
  __shared__ float tile[3][64][65];
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 64;
  for (int r = rd_ty(); r < 64; r += 4) {
    const int x = x0 + threadIdx.x, y = y0 + r;
    if (x < W && y < H) {
      const size_t i = (size_t)y * W + x;
      if (MODE == 1) {
        float l, a, b;
        unpack_lab(plab[i], l, a, b);
        tile[0][r][threadIdx.x] = l; tile[1][r][threadIdx.x] = a; tile[2][r][threadIdx.x] = b;
      } else if (MODE == 2) {
        // iu:580-589: horizontal result = anti-causal + causal - c0 * input
        for (int k = 0; k < np; k++) tile[k][r][threadIdx.x] = bwd.p[k][i] + fwd.p[k][i] - src.p[k][i] * 0.3989422804f;
      } else {
        for (int k = 0; k < np; k++) tile[k][r][threadIdx.x] = src.p[k][i];
      }
    }
  }
  __syncthreads();
  for (int r = rd_ty(); r < 64; r += 4) {
    const int ox = y0 + threadIdx.x, oy = x0 + r;   // output plane is H wide, W tall
    if (ox < H && oy < W)
      for (int k = 0; k < np; k++) dst.p[k][(size_t)oy * H + ox] = tile[k][threadIdx.x][r];
  }
}

// ------------------------------------------------------------------------------------------------ IIR Gaussian
// iu:542-627 with iircoef[2] (iu:915-921): d[x] = c0 in[x] + sum_{k=1..7} c_k in[x-k] + sum_{k=0..6} c_{8+k} d[x-1-k],
// evaluated in exactly that association; the sweep starts 11 samples before the line with mirrored inputs and zero state.
// One lane owns one column of a W-column plane and walks down (dir 0) or up (dir 1) its H samples; lanes of a wave
// read/write consecutive addresses.  The recurrence is latency bound (8 dependent ops per step), so loads are
// software-pipelined one chunk ahead.
#define IIR_C0 0.3989422804f
#define IIR_C1 0.1414542400f
#define IIR_C2 -0.0030406818f
#define IIR_C3 -0.0041116157f
#define IIR_C4 0.0006696623f
#define IIR_C5 0.0000498707f
#define IIR_C6 -0.0000449761f
#define IIR_C7 -0.0000051528f
#define IIR_C8 0.2519574622f
#define IIR_C9 -0.0098627835f
#define IIR_C10 -0.0067013653f
#define IIR_C11 0.0012572396f
#define IIR_C12 0.0000481394f
#define IIR_C13 -0.0000097781f
#define IIR_C14 0.0000006462f
#define IIR_WARM 11
#ifndef IIR_CH
#define IIR_CH 16               // rows per chunk of the interior path (loads of the next chunk in flight while one is evaluated)
#endif

// Both sweeps and the combination (iu:580-589 / iu:629-637: anti-causal + causal - c0 * input) for one 64-column x IF_ROWS
// block of one plane in ONE wave: the causal outputs of the block wait in LDS while the anti-causal sweep runs over the same
// rows and finishes each pixel, so neither sweep's result travels through HBM.  Both sweeps start IF_WU rows outside the
// block from a zero state (or at the true beginning of the sweep when that is nearer); whether that reproduces the full
// sweep BIT FOR BIT is checked on the device: each block records the 7 outputs it computed just before entering its rows
// ("warm") and its own last 7 outputs ("true"), k_iir_check_fix compares neighbours and evaluates a column with any
// difference again by full-length sweeps (rdk::iir_blur_pass).  TOUT = 1 writes the result transposed (through the LDS tile).
#define IF_ROWS_MIN 64         // rows per block: 64 or 128, chosen per launch (if_pick_rows)
                              // (a remainder of fewer than 8 rows is merged into the last block: LDS tile = rows + 8)
#define IF_WU 32              // warm-up rows (24 sufficed on every plane tried on the CPU; the on-device check is what guarantees the result)
#define IF_PITCH 65
__host__ __device__ inline int if_nchunks(int H, int rows) {
  int n = (H + rows - 1) / rows;
  if (n > 1 && H % rows != 0 && H % rows < 8) n--;
  return n;
}

// One step of the recurrence (iu:551-556): d = i0 c0; d += c1 i1 + ... + c7 i7; d += c8 t0 + ... + c14 t6 - each of the two sums from left
// to right.  The two sums are independent chains of seven products, so they run side by side in the two halves of packed single-precision
// operations (v_pk_mul_f32 / v_pk_add_f32: the same IEEE multiply and add per half, not fused): the state is kept as pairs
// q_k = (input k steps back, output k steps back), which is also the pair the k-th tap of each sum needs.
typedef float iir_f2 __attribute__((ext_vector_type(2)));
#define IIR_STATE iir_f2 q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0, q6 = 0, q7 = 0
#define IIR_STEP(i0v)                                                                                                  \
  iir_f2 acc = (iir_f2){IIR_C1, IIR_C8} * q1;                                                                          \
  acc = acc + (iir_f2){IIR_C2, IIR_C9} * q2;                                                                           \
  acc = acc + (iir_f2){IIR_C3, IIR_C10} * q3;                                                                          \
  acc = acc + (iir_f2){IIR_C4, IIR_C11} * q4;                                                                          \
  acc = acc + (iir_f2){IIR_C5, IIR_C12} * q5;                                                                          \
  acc = acc + (iir_f2){IIR_C6, IIR_C13} * q6;                                                                          \
  acc = acc + (iir_f2){IIR_C7, IIR_C14} * q7;                                                                          \
  float d = (i0v) * IIR_C0;                                                                                            \
  d += acc.x;                                                                                                          \
  d += acc.y;
#define IIR_SHIFT(i0v)                                                                                                 \
  q7 = q6; q6 = q5; q5 = q4; q4 = q3; q3 = q2; q2 = q1; q1 = (iir_f2){(i0v), d};

// TWO waves per block: wave 0 runs the causal sweep, wave 1 the anti-causal sweep of the same 64 columns AT THE SAME TIME.  Each parks
// its raw outputs for the half of the block it reaches first in LDS; after one barrier in the middle each finishes the other half
// with what the other wave parked there (anti-causal + causal - c0 * input: the sum commutes, so who adds does not matter).  The
// recurrence is a chain of dependent operations - a wave's time is its step count - so the block takes rows + run-in steps instead
// of twice that.
// SRC16: the source planes hold 16-bit integer fields of a packed Lab value (k_bgr2plab_t); plane 0 = L (12 bits), 1 and 2 = a, b
// (10 bits).  iir_field turns one into the float unpack_lab (iu:36-39) makes of it: a multiply and an add, not fused.
template <int SRC16> struct iir_src { typedef float T; };
template <> struct iir_src<1> { typedef unsigned short T; };
template <int SRC16, typename T> __device__ __forceinline__ float iir_field(T v, float sc, float hf) {
  if (SRC16) return (float)(int)v * sc + hf;
  return (float)v;
}
template <int TOUT, int IF_ROWS, int SRC16>
__global__ __launch_bounds__(128) void k_iir_fused(P3 dst, P3c src, float *__restrict__ tails, int W, int H, int nchunks, int *bad, int np, size_t zs) {
  __shared__ float fwt[(IF_ROWS + 8) * IF_PITCH];
  const int lane = threadIdx.x & 63, anti = threadIdx.x >> 6;
  // (grid: x = 64-column strips, y = plane + np * frame of a group launch, z = blocks of rows)
  const int k = blockIdx.y % np, c = blockIdx.z;
  // The plane's pointers picked by comparisons, not by indexing the structs: an index that is not a constant sends the struct through memory, the pointers come
  // back without their address space and every access becomes a flat instruction with a 64-bit address per lane.  Rows are then addressed as a UNIFORM row pointer
  // (scalar registers, advanced by scalar additions) plus the lane's 32-bit byte offset - no vector instruction per access.
  float *dk = k == 0 ? dst.p[0] : (k == 1 ? dst.p[1] : dst.p[2]);
  const float *sk = k == 0 ? src.p[0] : (k == 1 ? src.p[1] : src.p[2]);
  { const size_t rd_zoff_ = (size_t)(blockIdx.y / np) * zs; RD_ZS1(dk); RD_ZS1(sk); RD_ZS1(tails); RD_ZS1(bad); }
  if (blockIdx.x == 0 && k == 0 && blockIdx.z == 0 && threadIdx.x == 0) *bad = 0;      // diagnostics flag of the check that follows this launch
  const int x = blockIdx.x * 64 + lane;
  const bool xin = x < W;
  typedef typename iir_src<SRC16>::T TS;
  const TS *__restrict__ in = (const TS *)sk;                              // (uniform)
  const unsigned xs = (unsigned)(xin ? x : W - 1) * (unsigned)sizeof(TS), x4 = (unsigned)(xin ? x : W - 1) * 4u;      // the lane's byte offset within a row
#define IIR_ROW(ptr) (*(const TS *)((const char *)(ptr) + xs))               /* the lane's element of the source row that starts at `ptr` */
#define IIR_OUT(ptr) (*(float *)((char *)(ptr) + x4))                        /* the lane's element of a row of floats */
  const float sc = k == 0 ? 1.0f / 4096 : 1.0f / 1024, hf = k == 0 ? 0.5f / 4096 : 0.5f / 1024;
#define IIR_LD(v) iir_field<SRC16>(v, sc, hf)
  const int s0 = c * IF_ROWS, s1 = (c == nchunks - 1) ? H : s0 + IF_ROWS;
  const int mid = s0 + (s1 - s0) / 2;             // rows [s0, mid) are finished by the anti-causal wave, [mid, s1) by the causal one
  // tails: [plane][chunk][set: 0 fwd warm, 1 fwd true, 2 bwd warm, 3 bwd true][7][W]
  float *__restrict__ tl = tails + ((size_t)(k * nchunks + c) * 4 * 7) * W;      // (uniform)
  const int ylo = -IIR_WARM, yhi = H + IIR_WARM;
  float cur[IIR_CH], nxt[IIR_CH];
  static_assert(IIR_CH >= 8 && IF_WU % IIR_CH == 0 && (IF_ROWS / 2) % IIR_CH == 0, "the interior path walks whole chunks, half a block per phase");
  if (s0 - IF_WU >= 0 && s1 + IF_WU <= H && s1 - s0 == IF_ROWS) {
    // Interior block (all but the first and last of a column): no mirrored rows, no clamping, a full block - the same steps
    // as below with every row test resolved at compile time (the scalar address and branch work of the general form costs
    // as many issue slots as the recurrence itself).  A chunk = IIR_CH rows; while one is evaluated the next one's rows are
    // fetched (MORE = 0: there is no next one).
    // IIR_CHUNK(row pointer of the chunk's first row, row step, STORE: what to do with an output row, TAIL: tails set or -1)
#define IIR_CHUNK(PTR, STEP, STORE, TAIL, TIDX) IIR_CHUNK_(PTR, STEP, STORE, TAIL, TIDX, 1)
#define IIR_CHUNK_(PTR, STEP, STORE, TAIL, TIDX, MORE)                                                                  \
    {                                                                                                                    \
      if (MORE) { _Pragma("unroll") for (int j = 0; j < IIR_CH; j++) nxt[j] = IIR_LD(IIR_ROW((PTR) + ((long)(IIR_CH + j) * (STEP)) * W)); } \
      _Pragma("unroll") for (int j = 0; j < IIR_CH; j++) {                                                               \
        IIR_STEP(cur[j]);                                                                                                \
        STORE;                                                                                                           \
        if ((TAIL) >= 0 && j >= IIR_CH - 7 && xin) IIR_OUT(tl + (size_t)((TAIL) * 7 + (TIDX)) * W) = d;                            \
        IIR_SHIFT(cur[j]);                                                                                               \
      }                                                                                                                  \
      _Pragma("unroll") for (int j = 0; j < IIR_CH; j++) cur[j] = nxt[j];                                                \
    }
    constexpr int HALF = IF_ROWS / 2 / IIR_CH;     // chunks per phase
    IIR_STATE;
    if (!anti) {   // causal: rows s0 - IF_WU .. s1 - 1
      const TS *p = in + (size_t)(s0 - IF_WU) * W;
#pragma unroll
      for (int j = 0; j < IIR_CH; j++) cur[j] = IIR_LD(IIR_ROW(p + (size_t)j * W));
      for (int q = 0; q < IF_WU / IIR_CH - 1; q++) { IIR_CHUNK(p, 1, (void)0, -1, 0); p += (size_t)IIR_CH * W; }
      IIR_CHUNK(p, 1, (void)0, 0, j - (IIR_CH - 7)); p += (size_t)IIR_CH * W;                    // rows s0-16 .. s0-1: "warm" tails
      float *f = fwt + lane;
      for (int q = 0; q < HALF; q++) { IIR_CHUNK(p, 1, f[j * IF_PITCH] = d, -1, 0); p += (size_t)IIR_CH * W; f += IIR_CH * IF_PITCH; }
      __syncthreads();
      float *o = TOUT ? nullptr : dk + (size_t)mid * W;
#define IIR_FINISH_F { const float r = (f[j * IF_PITCH] + d) - cur[j] * IIR_C0; if (TOUT) f[j * IF_PITCH] = r; else if (xin) IIR_OUT(o + (long)j * W) = r; }
      for (int q = 0; q < HALF - 1; q++) { IIR_CHUNK(p, 1, IIR_FINISH_F, -1, 0); p += (size_t)IIR_CH * W; f += IIR_CH * IF_PITCH; if (!TOUT) o += (size_t)IIR_CH * W; }
      IIR_CHUNK_(p, 1, IIR_FINISH_F, 1, j - (IIR_CH - 7), 0);                                     // rows s1-16 .. s1-1: "true" tails
#undef IIR_FINISH_F
    } else {       // anti-causal: rows s1 - 1 + IF_WU .. s0
      const TS *p = in + (size_t)(s1 - 1 + IF_WU) * W;
#pragma unroll
      for (int j = 0; j < IIR_CH; j++) cur[j] = IIR_LD(IIR_ROW(p - (long)j * W));
      for (int q = 0; q < IF_WU / IIR_CH - 1; q++) { IIR_CHUNK(p, -1, (void)0, -1, 0); p -= (size_t)IIR_CH * W; }
      IIR_CHUNK(p, -1, (void)0, 2, (IIR_CH - 1) - j); p -= (size_t)IIR_CH * W;                   // rows s1+15 .. s1: "warm" tails (index = row - s1)
      float *f = fwt + (IF_ROWS - 1) * IF_PITCH + lane;
      for (int q = 0; q < HALF; q++) { IIR_CHUNK(p, -1, f[-j * IF_PITCH] = d, -1, 0); p -= (size_t)IIR_CH * W; f -= IIR_CH * IF_PITCH; }
      __syncthreads();
      float *o = TOUT ? nullptr : dk + (size_t)(mid - 1) * W;
#define IIR_FINISH_B { const float r = d + f[-j * IF_PITCH] - cur[j] * IIR_C0; if (TOUT) f[-j * IF_PITCH] = r; else if (xin) IIR_OUT(o - (long)j * W) = r; }
      for (int q = 0; q < HALF - 1; q++) { IIR_CHUNK(p, -1, IIR_FINISH_B, -1, 0); p -= (size_t)IIR_CH * W; f -= IIR_CH * IF_PITCH; if (!TOUT) o -= (size_t)IIR_CH * W; }
      IIR_CHUNK_(p, -1, IIR_FINISH_B, 3, (IIR_CH - 1) - j, 0);                                     // rows s0+15 .. s0: "true" tails (index = row - s0)
#undef IIR_FINISH_B
    }
#undef IIR_CHUNK
#undef IIR_CHUNK_
  } else {
    IIR_STATE;
    if (!anti) {   // ---------------- causal sweep: rows fb .. s1-1 (one barrier, before row `mid`)
      const int fb = (s0 - IF_WU <= ylo) ? ylo : s0 - IF_WU;
      const int total = s1 - fb;
#pragma unroll
      for (int j = 0; j < IIR_CH; j++) cur[j] = IIR_LD(IIR_ROW(in + (size_t)mirror1(clampi(fb + j, ylo, yhi), H) * W));
      for (int base = 0; base < total; base += IIR_CH) {
#pragma unroll
        for (int j = 0; j < IIR_CH; j++) nxt[j] = IIR_LD(IIR_ROW(in + (size_t)mirror1(clampi(fb + base + IIR_CH + j, ylo, yhi), H) * W));
#pragma unroll
        for (int j = 0; j < IIR_CH; j++) {
          const int yy = fb + base + j;
          if (yy == mid) __syncthreads();
          IIR_STEP(cur[j]);
          if (yy >= s0 && yy < mid) fwt[(yy - s0) * IF_PITCH + lane] = d;
          else if (yy >= mid && yy < s1) {
            const float o = (fwt[(yy - s0) * IF_PITCH + lane] + d) - cur[j] * IIR_C0;
            if (TOUT) fwt[(yy - s0) * IF_PITCH + lane] = o;
            else if (xin) IIR_OUT(dk + (size_t)yy * W) = o;
          }
          if (xin && yy >= s0 - 7 && yy < s0) IIR_OUT(tl + (size_t)(0 * 7 + yy - (s0 - 7)) * W) = d;
          if (xin && yy >= s1 - 7 && yy < s1) IIR_OUT(tl + (size_t)(1 * 7 + yy - (s1 - 7)) * W) = d;
          IIR_SHIFT(cur[j]);
        }
#pragma unroll
        for (int j = 0; j < IIR_CH; j++) cur[j] = nxt[j];
      }
    } else {       // ---------------- anti-causal sweep: rows bb .. s0, descending (one barrier, before row `mid - 1`)
      const int bb = (s1 - 1 + IF_WU >= yhi) ? yhi : s1 - 1 + IF_WU;
      const int total = bb - s0 + 1;
#pragma unroll
      for (int j = 0; j < IIR_CH; j++) cur[j] = IIR_LD(IIR_ROW(in + (size_t)mirror1(clampi(bb - j, ylo, yhi), H) * W));
      for (int base = 0; base < total; base += IIR_CH) {
#pragma unroll
        for (int j = 0; j < IIR_CH; j++) nxt[j] = IIR_LD(IIR_ROW(in + (size_t)mirror1(clampi(bb - (base + IIR_CH + j), ylo, yhi), H) * W));
#pragma unroll
        for (int j = 0; j < IIR_CH; j++) {
          const int yy = bb - (base + j);
          if (yy == mid - 1) __syncthreads();
          IIR_STEP(cur[j]);
          if (yy >= mid && yy < s1) fwt[(yy - s0) * IF_PITCH + lane] = d;
          else if (yy >= s0 && yy < mid) {
            const float o = d + fwt[(yy - s0) * IF_PITCH + lane] - cur[j] * IIR_C0;
            if (TOUT) fwt[(yy - s0) * IF_PITCH + lane] = o;
            else if (xin) IIR_OUT(dk + (size_t)yy * W) = o;
          }
          if (xin && yy >= s1 && yy < s1 + 7) IIR_OUT(tl + (size_t)(2 * 7 + yy - s1) * W) = d;
          if (xin && yy >= s0 && yy < s0 + 7) IIR_OUT(tl + (size_t)(3 * 7 + yy - s0) * W) = d;
          IIR_SHIFT(cur[j]);
        }
#pragma unroll
        for (int j = 0; j < IIR_CH; j++) cur[j] = nxt[j];
      }
    }
  }
  if (TOUT) {
    __syncthreads();
    const int rows = s1 - s0;
    for (int col = anti; col < 64; col += 2) {
      const int xx = blockIdx.x * 64 + col;
      if (xx >= W) break;
      float *__restrict__ o = dk + (size_t)xx * H + s0;
      for (int r = lane; r < rows; r += 64) at32(o, (unsigned)r) = fwt[r * IF_PITCH + col];
    }
  }
}

#undef IIR_LD
#undef IIR_ROW
#undef IIR_OUT

// The state a block reached after its warm-up must equal, bit for bit, what its neighbour computed for the same rows.  One
// wave per (64 columns, plane, chunk) compares the chunk's two borders; a column with a difference - none has been seen, the
// check is what guarantees the result - is evaluated again by the full-length sweeps (iu:580-589 / iu:629-637, mirrored ends,
// IIR_WARM rows of run-in) by the lane that found it, through the `fwd` scratch plane (two chunks of one column may both do
// that: they write the same values).  `force` (diagnostics) treats every column of chunk 0 as different.
#define IC_CH 4
// EAGER: all 28 values of a lane are requested before the first comparison (one trip to memory instead of up to fourteen dependent ones: a single frame's
// launch is nothing but that latency - 35 -> 8 us at 1920x1080; in group launches, eight frames' worth of bytes, the short-circuit form is faster)
template <int TOUT, int SRC16, int EAGER>
__global__ __launch_bounds__(64 * IC_CH) void k_iir_check_fix(P3 dst, P3c src, P3 fwd, const float *__restrict__ tails, int *bad, int W, int H, int nchunks, int IF_ROWS, int force, int np, size_t zs) {
  const int x = blockIdx.x * 64 + threadIdx.x;
  const int k = blockIdx.y % np, c = blockIdx.z * IC_CH + rd_ty();      // (one wave per chunk, IC_CH chunks per block: a block per wave was 6000 dispatches per plane set)
  if (c >= nchunks) return;
  // (the plane's pointers by comparisons: k_iir_fused)
  float *dk = k == 0 ? dst.p[0] : (k == 1 ? dst.p[1] : dst.p[2]), *fk = k == 0 ? fwd.p[0] : (k == 1 ? fwd.p[1] : fwd.p[2]);
  const float *sk = k == 0 ? src.p[0] : (k == 1 ? src.p[1] : src.p[2]);
  { const size_t rd_zoff_ = (size_t)(blockIdx.y / np) * zs; RD_ZS1(dk); RD_ZS1(sk); RD_ZS1(fk); RD_ZS1(tails); RD_ZS1(bad); }
  if (x >= W) return;
  const int s0 = c * IF_ROWS, s1 = (c == nchunks - 1) ? H : s0 + IF_ROWS;
  const float *me = tails + ((size_t)(k * nchunks + c) * 4 * 7) * W + x;
  bool differ = force != 0 && c == 0;
  const bool chk_f = c > 0 && s0 - IF_WU > -IIR_WARM;                 // causal: my warm rows s0-7..s0-1 against the previous block's last rows
  const bool chk_b = c < nchunks - 1 && s1 - 1 + IF_WU < H + IIR_WARM;   // anti-causal: my warm rows s1..s1+6 against the next block's first rows
  const float *pv = tails + ((size_t)(k * nchunks + (chk_f ? c - 1 : c)) * 4 * 7) * W + x;
  const float *nx = tails + ((size_t)(k * nchunks + (chk_b ? c + 1 : c)) * 4 * 7) * W + x;
  if (EAGER) {
    unsigned a[14], b[14];
#pragma unroll
    for (int j = 0; j < 7; j++) {
      a[j] = __float_as_uint(me[(size_t)(0 * 7 + j) * W]); b[j] = __float_as_uint(pv[(size_t)(1 * 7 + j) * W]);
      a[7 + j] = __float_as_uint(me[(size_t)(2 * 7 + j) * W]); b[7 + j] = __float_as_uint(nx[(size_t)(3 * 7 + j) * W]);
    }
    unsigned df = 0, db = 0;
#pragma unroll
    for (int j = 0; j < 7; j++) { df |= a[j] ^ b[j]; db |= a[7 + j] ^ b[7 + j]; }
    differ = differ || (chk_f && df != 0) || (chk_b && db != 0);
  } else {
    if (chk_f) {
#pragma unroll
      for (int j = 0; j < 7; j++) differ = differ || (__float_as_uint(me[(size_t)(0 * 7 + j) * W]) != __float_as_uint(pv[(size_t)(1 * 7 + j) * W]));
    }
    if (chk_b) {
#pragma unroll
      for (int j = 0; j < 7; j++) differ = differ || (__float_as_uint(me[(size_t)(2 * 7 + j) * W]) != __float_as_uint(nx[(size_t)(3 * 7 + j) * W]));
    }
  }
  if (!differ) return;
  atomicOr(bad, 1);
  typedef typename iir_src<SRC16>::T TS;
  const TS *__restrict__ in = (const TS *)sk + x;
  const float sc = k == 0 ? 1.0f / 4096 : 1.0f / 1024, hf = k == 0 ? 0.5f / 4096 : 0.5f / 1024;
  float *fw = fk + x;
  {
    IIR_STATE;
    for (int yy = -IIR_WARM; yy < H; yy++) {
      const float i0 = iir_field<SRC16>(in[(size_t)mirror1(yy, H) * W], sc, hf);
      IIR_STEP(i0);
      if (yy >= 0) fw[(size_t)yy * W] = d;
      IIR_SHIFT(i0);
    }
  }
  {
    IIR_STATE;
    for (int yy = H + IIR_WARM; yy >= 0; yy--) {
      const float i0 = iir_field<SRC16>(in[(size_t)mirror1(yy, H) * W], sc, hf);
      IIR_STEP(i0);
      if (yy < H) {
        const float o = d + fw[(size_t)yy * W] - i0 * IIR_C0;
        if (TOUT) dk[(size_t)x * H + yy] = o; else dk[(size_t)yy * W + x] = o;
      }
      IIR_SHIFT(i0);
    }
  }
}

// ------------------------------------------------------------------------------------------------ IIR Gaussian, any radius
// oclimgutil_iirblur_f_f with r != 2 (no caller in the reference passes one; sigma = (r + 1) / 3, coefficients iu:900-1125).
// The blocked evaluation above rests on the sigma = 1 filter forgetting its start within a few dozen samples, bit for bit; wider
// filters do not, so these radii take the reference's own shape: every line swept over its full length (iu:542-627), the causal
// sweep by one thread and the anti-causal sweep by its neighbour in rd_ty(), each into its scratch plane, then combined.
#define RD_IIRCOEF_ATTR __device__ __constant__
#include "rd_iircoef.h"
__global__ __launch_bounds__(128) void k_iir_line(float *dst, const float *__restrict__ src, float *fw, float *bw, int W, int H, int r) {
  const int x = blockIdx.x * 64 + threadIdx.x;
  const bool xin = x < W;
  float C[15];
#pragma unroll
  for (int k = 0; k < 15; k++) C[k] = rd_iircoef[r][k];
  const int warm = r + 1 + 8;
  if (xin) {
    const float *in = src + x;
    float i1 = 0, i2 = 0, i3 = 0, i4 = 0, i5 = 0, i6 = 0, i7 = 0;
    float t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0;
#define IIR_LINE_STEP                                                                                                    \
    const float i0 = in[(size_t)mirror1(yy, H) * W];                                                                     \
    float d = i0 * C[0];                                                                                                 \
    d += C[1] * i1 + C[2] * i2 + C[3] * i3 + C[4] * i4 + C[5] * i5 + C[6] * i6 + C[7] * i7;                              \
    d += C[8] * t0 + C[9] * t1 + C[10] * t2 + C[11] * t3 + C[12] * t4 + C[13] * t5 + C[14] * t6;                         \
    i7 = i6; i6 = i5; i5 = i4; i4 = i3; i3 = i2; i2 = i1; i1 = i0;                                                       \
    t6 = t5; t5 = t4; t4 = t3; t3 = t2; t2 = t1; t1 = t0; t0 = d;
    if (rd_ty() == 0) {
      float *o = fw + x;
      for (int yy = -warm; yy < H; yy++) { IIR_LINE_STEP; if (yy >= 0) o[(size_t)yy * W] = d; }
    } else {
      float *o = bw + x;
      for (int yy = H + warm; yy >= 0; yy--) { IIR_LINE_STEP; if (yy < H) o[(size_t)yy * W] = d; }
    }
#undef IIR_LINE_STEP
  }
  __syncthreads();
  if (!xin) return;
  for (int yy = rd_ty(); yy < H; yy += 2) {       // iu:580-589 / iu:629-637: tmp1 + tmp0 - in * c0
    const size_t p = (size_t)yy * W + x;
    dst[p] = bw[p] + fw[p] - src[p] * C[0];
  }
}

// ------------------------------------------------------------------------------------------------ gradient direction
#include "rd_front_helpers.h"

// (pack_out, optional: the packed Lab of the pixel's own blurred L, a, b - iu:28-34 - on the way: saves the frame path a launch)
// Every thread handles EV_PX pixels below one another: their 5x5 windows overlap in all but one row each, so a column of four
// costs 8 x 5 loads instead of 100 (same sums in the same order per pixel).
#ifndef EV_PX
#define EV_PX 4
#endif
__device__ __forceinline__ void ev_finish(float2 *__restrict__ dst, int p, float vx, float vy) {
  float len = vx * vx + vy * vy;
  if ((double)len > 1e-10) {
    len = 1.0f / sqrtf(len);
    vx *= len; vy *= len;
  } else {
    vx = vy = 0.70710678118f;
  }
  dst[p] = make_float2(vx, vy);
}
__global__ __launch_bounds__(256) void k_edgevec(float2 *__restrict__ dst, const float *__restrict__ in, int iw, int ih, uint32_t *__restrict__ pack_out, const float *__restrict__ pa, const float *__restrict__ pb, size_t zs) {
  RD_ZSHIFT(zs, dst, in, pack_out, pa, pb);
  const int x = blockIdx.x * 64 + threadIdx.x, y0 = (blockIdx.y * 4 + rd_ty()) * EV_PX;
  if (x >= iw || y0 >= ih) return;
  if (pack_out != nullptr) {
#pragma unroll
    for (int k = 0; k < EV_PX; k++) if (y0 + k < ih) { const int p = (y0 + k) * iw + x; pack_out[p] = pack_lab(in[p], pa[p], pb[p]); }
  }
  // (blocks whose 5x5 windows stay inside the frame address their samples by constant offsets from the row pointers; same sums in the same order)
  const bool interior = blockIdx.x > 0 && blockIdx.y > 0 && (int)(blockIdx.x * 64 + 66) <= iw && (int)((blockIdx.y * 4 + 4) * EV_PX + 2) <= ih;
  if (interior) {
    float w[EV_PX + 4][5];
#pragma unroll
    for (int r = 0; r < EV_PX + 4; r++) {
      const float *row = in + (size_t)(y0 + r - 2) * iw + x;
#pragma unroll
      for (int c = 0; c < 5; c++) w[r][c] = row[c - 2];
    }
#pragma unroll
    for (int k = 0; k < EV_PX; k++) {
      float vx = 0, vy = 0;
#pragma unroll
      for (int yy = -2; yy <= 2; yy++)
#pragma unroll
        for (int xx = -2; xx <= 2; xx++) {
          const float s = w[k + yy + 2][xx + 2];
          vx += v5c((xx + 2) + (yy + 2) * 5) * s;
          vy += v5c((yy + 2) + (xx + 2) * 5) * s;
        }
      ev_finish(dst, (y0 + k) * iw + x, vx, vy);
    }
  } else {
    for (int k = 0; k < EV_PX; k++) {
      const int y = y0 + k;
      if (y >= ih) break;
      float vx = 0, vy = 0;
#pragma unroll
      for (int yy = -2; yy <= 2; yy++) {
#pragma unroll
        for (int xx = -2; xx <= 2; xx++) {
          const float s = in[mirror2(x + xx, y + yy, iw, ih)];
          vx += v5c((xx + 2) + (yy + 2) * 5) * s;
          vy += v5c((yy + 2) + (xx + 2) * 5) * s;
        }
      }
      ev_finish(dst, y * iw + x, vx, vy);
    }
  }
}

// ------------------------------------------------------------------------------------------------ edge strength
// iu:422-437 on the blurred packed Lab: per channel (NW-SE)(N+W-S-E) + (NE-SW)(N-W+E-S), clamped at 0, summed, sqrt
// (blocks whose 3x3 windows stay inside the frame - all but the frame's rim - address their neighbours by constant offsets: the
//  mirrored-index arithmetic of the general form costs as many instructions as the gradient itself)
// (every thread handles EP_PX pixels below one another: a column of four unpacks 6 x 3 cells instead of 32)
#ifndef EP_PX
#define EP_PX 4
#endif
__global__ __launch_bounds__(256) void k_edge_plab(float *__restrict__ out, const uint32_t *__restrict__ in, int iw, int ih, size_t zs) {
  RD_ZSHIFT(zs, out, in);
  const int x = blockIdx.x * 64 + threadIdx.x, y0 = (blockIdx.y * 4 + rd_ty()) * EP_PX;
  if (x >= iw || y0 >= ih) return;
  const bool interior = blockIdx.x > 0 && blockIdx.y > 0 && (int)(blockIdx.x * 64 + 64) < iw && (int)((blockIdx.y * 4 + 4) * EP_PX) < ih;
  if (interior) {
    float u[EP_PX + 2][3][3];       // [row][column][channel]
#pragma unroll
    for (int r = 0; r < EP_PX + 2; r++) {
      const uint32_t *c = in + (size_t)(y0 + r - 1) * iw + x;
#pragma unroll
      for (int k = 0; k < 3; k++) unpack_lab(c[k - 1], u[r][k][0], u[r][k][1], u[r][k][2]);
    }
#pragma unroll
    for (int k = 0; k < EP_PX; k++)
      out[(y0 + k) * iw + x] = ep_strength(u[k][1], u[k + 2][1], u[k + 1][0], u[k + 1][2], u[k][0], u[k][2], u[k + 2][0], u[k + 2][2]);
    return;
  }
  for (int k = 0; k < EP_PX; k++) {
    const int y = y0 + k;
    if (y >= ih) break;
    float n[3], s[3], w[3], e[3], nw[3], ne[3], sw[3], se[3];
    unpack_lab(in[mirror2(x, y - 1, iw, ih)], n[0], n[1], n[2]);
    unpack_lab(in[mirror2(x, y + 1, iw, ih)], s[0], s[1], s[2]);
    unpack_lab(in[mirror2(x - 1, y, iw, ih)], w[0], w[1], w[2]);
    unpack_lab(in[mirror2(x + 1, y, iw, ih)], e[0], e[1], e[2]);
    unpack_lab(in[mirror2(x - 1, y - 1, iw, ih)], nw[0], nw[1], nw[2]);
    unpack_lab(in[mirror2(x + 1, y - 1, iw, ih)], ne[0], ne[1], ne[2]);
    unpack_lab(in[mirror2(x - 1, y + 1, iw, ih)], sw[0], sw[1], sw[2]);
    unpack_lab(in[mirror2(x + 1, y + 1, iw, ih)], se[0], se[1], se[2]);
    out[y * iw + x] = ep_strength(n, s, w, e, nw, ne, sw, se);
  }
}

// ------------------------------------------------------------------------------------------------ non-max suppression
// iu:87-94 + iu:456-471: strength sampled at +-1, +-2 along the gradient direction with a 4x4 bicubic (mirrored
// borders); local maxima keep the 5-sample sum, everything else 0.
__device__ __forceinline__ float bicubic(const float *__restrict__ p, float x, float y, int iw, int ih) {
  const int ix = (int)x, iy = (int)y;
  const float fx = x - ix, fy = y - iy;
  const int xa = mirror1(ix - 1, iw), xb = mirror1(ix, iw), xc = mirror1(ix + 1, iw), xd = mirror1(ix + 2, iw);
  float r[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const float *row = p + (size_t)mirror1(iy - 1 + k, ih) * iw;
    r[k] = cubic1(row[xa], row[xb], row[xc], row[xd], fx);
  }
  return cubic1(r[0], r[1], r[2], r[3], fy);
}

// The four sample positions lie within 2 px of the pixel and the 4x4 footprints within [-3, +4]: the block stages a
// (64+8) x (TT_ROWS+7) tile of the strength plane in LDS with the mirrored border already applied, so a bicubic is one
// address computation and 16 ds_reads at constant offsets instead of 8 mirror clamps and 16 global gathers.  The outer
// two samples are only evaluated for local maxima (they do not influence the comparison).
#ifndef TT_ROWS
#define TT_ROWS 8
#endif

// Two bicubic samples (one step along the gradient direction either way) decide whether a pixel is a local maximum; only maxima -
// a third of the pixels of a noisy frame, scattered over every wave - need the two outer samples.  Evaluated inside the same
// loop, the outer samples would be computed by whole waves for the sake of a few lanes: the block collects its maxima in an LDS
// list instead and works the list off with all lanes busy.
__global__ __launch_bounds__(256) void k_thinthres(float *__restrict__ out, const float *__restrict__ in, const float2 *__restrict__ vxy, int iw, int ih, size_t zs) {
  RD_ZSHIFT(zs, out, in, vxy);
  __shared__ float tile[(TT_ROWS + 7) * TT_PITCH];
  __shared__ float4 lst[64 * TT_ROWS];          // a maximum: {its inner samples am1, ap1, its direction}
  __shared__ int lpix[64 * TT_ROWS];            // ... and its tile cell (row * 64 + column)
  __shared__ int nlst;
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * TT_ROWS;
  const int tid = rd_ty() * 64 + threadIdx.x;
  const int x = x0 + threadIdx.x;
  if (tid == 0) nlst = 0;
  // the directions of this thread's pixels are requested together with the tile (one wait for memory per block)
  float2 dir[TT_ROWS / 4];
#pragma unroll
  for (int k = 0; k < TT_ROWS / 4; k++) {
    const int y = y0 + rd_ty() + 4 * k;
    dir[k] = vxy[(x < iw && y < ih) ? y * iw + x : 0];
  }
  stage_cells<(TT_ROWS + 7) * TT_PITCH, 256>(tid, in,
    [&](int t, int &a) { a = mirror1(y0 - 3 + t / TT_PITCH, ih) * iw + mirror1(x0 - 3 + t % TT_PITCH, iw); return true; },
    [&](int t, bool, float v) { tile[t] = v; });
  __syncthreads();
#pragma unroll
  for (int k = 0; k < TT_ROWS / 4; k++) {
    const int r = rd_ty() + 4 * k;
    const int y = y0 + r;
    const bool inside = x < iw && y < ih;
    bool peak = false;
    float am1 = 0, ap1 = 0;
    const float2 v = dir[k];
    if (inside) {
      const float a0 = tile[(r + 3) * TT_PITCH + threadIdx.x + 3];
      am1 = bicubic_lds(tile, x - 1 * v.x, y - 1 * v.y, x0, y0);
      ap1 = bicubic_lds(tile, x + 1 * v.x, y + 1 * v.y, x0, y0);
      peak = am1 <= a0 && a0 >= ap1;
      if (!peak) out[y * iw + x] = 0.0f;
    }
    const unsigned long long m = __ballot(peak);
    if (m) {
      const int lane = threadIdx.x, leader = __ffsll((long long)m) - 1;
      int b = 0;
      if (lane == leader) b = atomicAdd(&nlst, __popcll(m));
      b = __shfl(b, leader);
      if (peak) { const int i = b + __popcll(m & ((1ull << lane) - 1)); lst[i] = make_float4(am1, ap1, v.x, v.y); lpix[i] = r * 64 + threadIdx.x; }
    }
  }
  __syncthreads();
  const int n = nlst;
  for (int i = tid; i < n; i += 256) {
    const float4 e = lst[i];
    const int c = lpix[i], r = c >> 6, tx = c & 63;
    const int xx = x0 + tx, yy = y0 + r;
    const float a0 = tile[(r + 3) * TT_PITCH + tx + 3];
    const float am2 = bicubic_lds(tile, xx - 2 * e.z, yy - 2 * e.w, x0, y0);
    const float ap2 = bicubic_lds(tile, xx + 2 * e.z, yy + 2 * e.w, x0, y0);
    out[yy * iw + xx] = am2 + e.x + a0 + e.y + ap2;
  }
}

// ------------------------------------------------------------------------------------------------ element-wise (iu:197-254)
__global__ void k_threshold_f(float *__restrict__ out, const float *__restrict__ in, float lo, float thr, float hi, int n) {
  ew4(n, [=](int i) { return in[i]; }, [=](int i, float v) { out[i] = v > thr ? hi : lo; });
}
__global__ void k_threshold_i(int *__restrict__ out, const int *__restrict__ in, int lo, int thr, int hi, int n) {
  ew4(n, [=](int i) { return in[i]; }, [=](int i, int v) { out[i] = v > thr ? hi : lo; });
}
__global__ void k_cast_i_f(int *__restrict__ out, const float *__restrict__ in, float scale, int n) {
  ew4(n, [=](int i) { return in[i]; }, [=](int i, float v) { out[i] = (int)(v * scale); });
}
__global__ void k_cast_c_i(int8_t *__restrict__ out, const int *__restrict__ in, int n) {
  ew4(n, [=](int i) { return in[i]; }, [=](int i, int v) { out[i] = (int8_t)v; });
}
__global__ void k_clear_i(int *out, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = 0;
}
__global__ void k_copy_i(int *out, const int *in, int n) {     // (in-place aliasing allowed: no restrict, plain loop)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void k_rand_i(int *out, uint64_t seed, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = pixel_rand(i, seed);
}

inline int ew_grid(int n) { int g = cdiv(n, 256 * 4); return g < 1 ? 1 : (g > 4096 ? 4096 : g); }

// ------------------------------------------------------------------------------------------------ visualisers and operators
// no application of the reference calls (oclimgutil.h:84-98): plain one-pixel-per-thread kernels, same operation order
// as the reference's (iu = oclimgutil.cl), pinned by tests/golden/ops_*.npz

// iu:283-289
__global__ __launch_bounds__(256) void k_convert_bgr_lumaf(uint8_t *__restrict__ out, const float *__restrict__ in, float f, int iw, int ih, int ws) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + rd_ty();
  if (x >= iw || y >= ih) return;
  const uint8_t v = (uint8_t)clampi((int)floorf(in[y * iw + x] * f * 255), 0, 255);
  uint8_t *o = out + (size_t)y * ws + x * 3;
  o[0] = v; o[1] = v; o[2] = v;
}

// iu:291-321
__global__ __launch_bounds__(256) void k_convert_bgr_labeli(uint8_t *__restrict__ out, const int *__restrict__ in, int bgc, int iw, int ih, int ws) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + rd_ty();
  if (x >= iw || y >= ih) return;
  const int c = in[y * iw + x];
  uint8_t *o = out + (size_t)y * ws + x * 3;
  if (c == bgc) { o[0] = 0; o[1] = 0; o[2] = 0; return; }
  const int g = (int)((unsigned)c * 1103515245u + 12345u);
  o[2] = (uint8_t)((((g & (7 << 0)) << 5) | 31) & 255);
  o[1] = (uint8_t)((((g & (7 << 3)) << 2) | 31) & 255);
  o[0] = (uint8_t)((((g & (7 << 6)) >> 1) | 31) & 255);
}

// iu:136-182, iu:264-273: packed Lab -> sRGB bytes
__device__ __forceinline__ float icfunc(float ft) { return ft > 0.20689270648f ? ft * ft * ft : (ft - 16.0f / 116) * (1.0f / 7.787f); }

__global__ __launch_bounds__(256) void k_plab2bgr(uint8_t *__restrict__ out, const uint32_t *__restrict__ in, int iw, int ih, int ws) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + rd_ty();
  if (x >= iw || y >= ih) return;
  const float xn = 0.950456f, zn = 1.088754f;
  float L, a, b;
  unpack_lab(in[y * iw + x], L, a, b);
  L *= 256; a *= 256; b *= 256;
  float cy;
  if (L > 0.20689270648f) { cy = (L + 16) * (1.0f / 116.0f); cy = cy * cy * cy; }
  else cy = L * (1.0f / 903.3f);
  const float fy = (float)(rd_lut_cfunc[clampi((int)floorf(cy * 1024), 0, 1023)] + 9039) * (1.0f / 65536.0f);
  const float fz = fy - (b - 128) * (1.0f / 200.0f);
  const float fx = fy + (a - 128) * (1.0f / 500.0f);
  const float cx = icfunc(fx) * xn, cz = icfunc(fz) * zn;
  const float r = cx * 3.240479f + cy * -1.537150f + cz * -0.498535f;
  const float g = cx * -0.969256f + cy * 1.875991f + cz * 0.041556f;
  const float bb = cx * 0.055648f + cy * -0.204043f + cz * 1.057311f;
  uint8_t *o = out + (size_t)y * ws + x * 3;
  o[2] = rd_lut_l2s[clampi((int)floorf(r * 1024), 0, 1023)];
  o[1] = rd_lut_l2s[clampi((int)floorf(g * 1024), 0, 1023)];
  o[0] = rd_lut_l2s[clampi((int)floorf(bb * 1024), 0, 1023)];
}

// iu:439-453
__global__ __launch_bounds__(256) void k_edge_f(float *__restrict__ out, const float *__restrict__ in, int iw, int ih) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + rd_ty();
  if (x >= iw || y >= ih) return;
  const float n = in[mirror2(x, y - 1, iw, ih)], s = in[mirror2(x, y + 1, iw, ih)], w = in[mirror2(x - 1, y, iw, ih)], e = in[mirror2(x + 1, y, iw, ih)];
  float sum = 0, t;
  t = n + w - s - e;
  sum += (in[mirror2(x - 1, y - 1, iw, ih)] - in[mirror2(x + 1, y + 1, iw, ih)]) * t;
  t = n - w + e - s;
  sum += (in[mirror2(x + 1, y - 1, iw, ih)] - in[mirror2(x - 1, y + 1, iw, ih)]) * t;
  out[y * iw + x] = sqrtf(fmaxf(0.0f, sum));
}

// iu:354-393: gradient direction of the channel with the largest response, sign taken from the L channel
__global__ __launch_bounds__(256) void k_edgevec_plab(float2 *__restrict__ dst, const uint32_t *__restrict__ in, int iw, int ih) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + rd_ty();
  if (x >= iw || y >= ih) return;
  float vx3[3] = { 0, 0, 0 }, vy3[3] = { 0, 0, 0 };
#pragma unroll
  for (int yy = -2; yy <= 2; yy++) {
#pragma unroll
    for (int xx = -2; xx <= 2; xx++) {
      float s[3];
      unpack_lab(in[mirror2(x + xx, y + yy, iw, ih)], s[0], s[1], s[2]);
#pragma unroll
      for (int c = 0; c < 3; c++) {
        vx3[c] += v5c((xx + 2) + (yy + 2) * 5) * s[c];
        vy3[c] += v5c((yy + 2) + (xx + 2) * 5) * s[c];
      }
    }
  }
  float l3[3];
#pragma unroll
  for (int c = 0; c < 3; c++) l3[c] = vx3[c] * vx3[c] + vy3[c] * vy3[c];
  float ivlen, vx, vy;
  if (l3[0] >= l3[1] && l3[0] >= l3[2]) { ivlen = l3[0]; vx = vx3[0]; vy = vy3[0]; }
  else if (l3[1] >= l3[2]) { ivlen = l3[1]; vx = vx3[1]; vy = vy3[1]; }
  else { ivlen = l3[2]; vx = vx3[2]; vy = vy3[2]; }
  if ((double)l3[0] >= 1e-6 && (vx3[0] * vx + vy3[0] * vy < 0)) { vx = -vx; vy = -vy; }
  if ((double)ivlen > 1e-10) { ivlen = 1.0f / sqrtf(ivlen); vx *= ivlen; vy *= ivlen; }
  else vx = vy = 0.70710678118f;
  dst[y * iw + x] = make_float2(vx, vy);
}

// iu:473-491: like k_thinthres, with a 1 % tolerance and all four samples in the comparison
__global__ __launch_bounds__(256) void k_thincubic(float *__restrict__ out, const float *__restrict__ in, const float2 *__restrict__ vxy, int iw, int ih) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + rd_ty();
  if (x >= iw || y >= ih) return;
  const int p0 = y * iw + x;
  const float2 v = vxy[p0];
  const float am2 = bicubic(in, x - 2 * v.x, y - 2 * v.y, iw, ih);
  const float am1 = bicubic(in, x - 1 * v.x, y - 1 * v.y, iw, ih);
  const float a0 = in[p0];
  const float ap1 = bicubic(in, x + 1 * v.x, y + 1 * v.y, iw, ih);
  const float ap2 = bicubic(in, x + 2 * v.x, y + 2 * v.y, iw, ih);
  const float C = 0.99f;
  out[p0] = (am2 * C <= a0 && am1 * C <= a0 && a0 >= ap1 * C && a0 >= ap2 * C) ? (am2 + am1 + a0 + ap1 + ap2) : 0.0f;
}

}  // namespace

namespace rdk {

void bgr2plab(hipStream_t s, uint32_t *out, const uint8_t *bgr, int iw, int ih, int ws) {
  hipLaunchKernelGGL(k_bgr2plab, dim3(cdiv(iw, 64), cdiv(ih, 16)), block2, 0, s, out, bgr, iw, ih, ws);
}
void bgr2plab_transposed(hipStream_t s, uint32_t *out, float *const dst[3], const uint8_t *const *bgr, int iw, int ih, int ws, int nz, size_t zs) {
  P3 d = { { dst[0], dst[1], dst[2] } };
  SrcZ z;
  for (int i = 0; i < RD_ZB_MAX; i++) z.p[i] = bgr[i < nz ? i : 0];
  hipLaunchKernelGGL(k_bgr2plab_t, dim3(cdiv(iw, 64), cdiv(ih, 64), nz), block2, 0, s, out, d, z, iw, ih, ws, zs);
}
void bgr2plab_transposed(hipStream_t s, uint32_t *out, float *const dst[3], const uint8_t *bgr, int iw, int ih, int ws) {
  bgr2plab_transposed(s, out, dst, &bgr, iw, ih, ws, 1, 0);
}
void unpack_plab(hipStream_t s, float *L, float *a, float *b, const uint32_t *in, int n) {
  hipLaunchKernelGGL(k_unpack_plab, dim3(ew_grid(n)), dim3(256), 0, s, L, a, b, in, n);
}
void pack_plab(hipStream_t s, uint32_t *out, const float *L, const float *a, const float *b, int n) {
  hipLaunchKernelGGL(k_pack_plab, dim3(ew_grid(n)), dim3(256), 0, s, out, L, a, b, n);
}

static P3 mk3(float *const p[3], int np) { P3 r = { { nullptr, nullptr, nullptr } }; for (int k = 0; k < np; k++) r.p[k] = p[k]; return r; }
static P3c mk3c(const float *const p[3], int np) { P3c r = { { nullptr, nullptr, nullptr } }; for (int k = 0; k < np; k++) r.p[k] = p[k]; return r; }

void transpose_f(hipStream_t s, float *const dst[3], const float *const src[3], int np, int W, int H) {
  P3c z = { { nullptr, nullptr, nullptr } };
  hipLaunchKernelGGL(k_transpose<0>, dim3(cdiv(W, 64), cdiv(H, 64)), block2, 0, s, mk3(dst, np), mk3c(src, np), z, z, (const uint32_t *)nullptr, np, W, H, (const int *)nullptr);
}
// Block height per launch.  One wave per block, each a serial recurrence of 2 x (rows + IF_WU) steps whose dependent chain is
// longer than its issue time: several waves per SIMD hide that, so short blocks win although they repeat the run-in more often
// (measured at 1080p, one wave per SIMD with 96 / 112 rows: 62 / 55 us; 32, 48, 64 rows: 48, 45, 50 us with transposed output,
// 37, 38, 40 us without).
static int if_pick_rows(int np, int W, int H, int transpose_out, int nz) {
  (void)np; (void)W; (void)H;
  (void)transpose_out;
  // (group launches: 128 rows - a quarter instead of half as much run-in, half as many borders to check; 2154-2167 against 2121-2151 frames/s
  //  in four interleaved pairs of runs; single frames keep 64: more blocks for the same device)
#ifndef IF_GROUP_ROWS
#define IF_GROUP_ROWS 128
#endif
  if (nz > 1) return IF_GROUP_ROWS;
  return 64;     // 32 rows of run-in per 64 rows of output: with several frames in flight the instruction count matters more than the wave count (48 / 32 rows ran 5 % slower there, though faster alone)
}

size_t iir_pass_scratch_floats(int np, int W, int H) { return (size_t)np * if_nchunks(H, IF_ROWS_MIN) * 4 * 7 * W; }   // (the finest blocking bounds all)

// one blur pass (both sweeps + combination) down the columns of np planes (W columns, H rows); transpose_out: dst planes
// are H wide, W tall.  fwd: scratch planes, only touched by columns whose blocked evaluation fails its on-device check and
// which are then evaluated by full-length sweeps (*bad is set to 1: diagnostics; bwd is not used any more); tails:
// iir_pass_scratch_floats() floats.
// src16 (only with transpose_out): the source planes hold 16-bit Lab fields (bgr2plab_transposed), plane 0 = L
void iir_blur_pass(hipStream_t s, float *const dst[3], const float *const src[3], float *const fwd[3], float *const bwd[3], int np, int W, int H,
                   int transpose_out, float *tails, int *bad, int src16, int nz, size_t zs) {
  const int rows = if_pick_rows(np, W, H, transpose_out, nz);
  const int nchunks = if_nchunks(H, rows);
  const dim3 grid(cdiv(W, 64), np * nz, nchunks);
#define IF_LAUNCH(T, R, S16) hipLaunchKernelGGL((k_iir_fused<T, R, S16>), grid, dim3(128), 0, s, mk3(dst, np), mk3c(src, np), tails, W, H, nchunks, bad, np, zs)
#define IF_LAUNCH_R(R) { if (transpose_out && src16) IF_LAUNCH(1, R, 1); else if (transpose_out) IF_LAUNCH(1, R, 0); else IF_LAUNCH(0, R, 0); }
  switch (rows) {
    case 64: IF_LAUNCH_R(64); break;
    default: IF_LAUNCH_R(128); break;
  }
#undef IF_LAUNCH_R
#undef IF_LAUNCH
  if (nchunks > 1) {
    const dim3 cgrid(grid.x, grid.y, cdiv(nchunks, IC_CH)), cblock(64, IC_CH);
    const int force = getenv("RD_IIR_FORCE_FIX") ? 1 : 0;             // diagnostics: every column takes the full-length path
#define IC_LAUNCH(T, S16, E) hipLaunchKernelGGL((k_iir_check_fix<T, S16, E>), cgrid, cblock, 0, s, mk3(dst, np), mk3c(src, np), mk3(fwd, np), (const float *)tails, bad, W, H, nchunks, rows, force, np, zs)
    if (nz > 1) { if (transpose_out && src16) IC_LAUNCH(1, 1, 0); else if (transpose_out) IC_LAUNCH(1, 0, 0); else IC_LAUNCH(0, 0, 0); }
    else { if (transpose_out && src16) IC_LAUNCH(1, 1, 1); else if (transpose_out) IC_LAUNCH(1, 0, 1); else IC_LAUNCH(0, 0, 1); }
#undef IC_LAUNCH
  }
  (void)bwd;
}

// one direction of the general-radius blur: lines run along y (W lines of H samples); dst may be fw or bw, not src
void iir_blur_lines(hipStream_t s, float *dst, const float *src, float *fw, float *bw, int W, int H, int r) {
  hipLaunchKernelGGL(k_iir_line, dim3(cdiv(W, 64)), dim3(64, 2), 0, s, dst, src, fw, bw, W, H, r);
}

void convert_bgr_lumaf(hipStream_t s, uint8_t *out, const float *in, float f, int iw, int ih, int ws) {
  hipLaunchKernelGGL(k_convert_bgr_lumaf, grid2(iw, ih), block2, 0, s, out, in, f, iw, ih, ws);
}
void convert_bgr_labeli(hipStream_t s, uint8_t *out, const int *in, int bgc, int iw, int ih, int ws) {
  hipLaunchKernelGGL(k_convert_bgr_labeli, grid2(iw, ih), block2, 0, s, out, in, bgc, iw, ih, ws);
}
void plab2bgr(hipStream_t s, uint8_t *out, const uint32_t *in, int iw, int ih, int ws) {
  hipLaunchKernelGGL(k_plab2bgr, grid2(iw, ih), block2, 0, s, out, in, iw, ih, ws);
}
void edge_f(hipStream_t s, float *out, const float *in, int iw, int ih) {
  hipLaunchKernelGGL(k_edge_f, grid2(iw, ih), block2, 0, s, out, in, iw, ih);
}
void edgevec_plab(hipStream_t s, float *vxy, const uint32_t *in, int iw, int ih) {
  hipLaunchKernelGGL(k_edgevec_plab, grid2(iw, ih), block2, 0, s, (float2 *)vxy, in, iw, ih);
}
void thincubic(hipStream_t s, float *out, const float *in, const float *vxy, int iw, int ih) {
  hipLaunchKernelGGL(k_thincubic, grid2(iw, ih), block2, 0, s, out, in, (const float2 *)vxy, iw, ih);
}
void edgevec(hipStream_t s, float *vxy, const float *in, int iw, int ih, uint32_t *pack_out, const float *a, const float *b, int nz, size_t zs) {
  hipLaunchKernelGGL(k_edgevec, dim3(cdiv(iw, 64), cdiv(ih, 4 * EV_PX), nz), block2, 0, s, (float2 *)vxy, in, iw, ih, pack_out, a, b, zs);
}
void edge_plab(hipStream_t s, float *out, const uint32_t *in, int iw, int ih, int nz, size_t zs) {
  hipLaunchKernelGGL(k_edge_plab, dim3(cdiv(iw, 64), cdiv(ih, 4 * EP_PX), nz), block2, 0, s, out, in, iw, ih, zs);
}
void thinthres(hipStream_t s, float *out, const float *in, const float *vxy, int iw, int ih, int nz, size_t zs) {
  hipLaunchKernelGGL(k_thinthres, dim3(cdiv(iw, 64), cdiv(ih, TT_ROWS), nz), dim3(64, 4), 0, s, out, in, (const float2 *)vxy, iw, ih, zs);
}
void threshold_f(hipStream_t s, float *out, const float *in, float lo, float thr, float hi, int n) {
  hipLaunchKernelGGL(k_threshold_f, dim3(ew_grid(n)), dim3(256), 0, s, out, in, lo, thr, hi, n);
}
void threshold_i(hipStream_t s, int *out, const int *in, int lo, int thr, int hi, int n) {
  hipLaunchKernelGGL(k_threshold_i, dim3(ew_grid(n)), dim3(256), 0, s, out, in, lo, thr, hi, n);
}
void cast_i_f(hipStream_t s, int *out, const float *in, float scale, int n) {
  hipLaunchKernelGGL(k_cast_i_f, dim3(ew_grid(n)), dim3(256), 0, s, out, in, scale, n);
}
void cast_c_i(hipStream_t s, int8_t *out, const int *in, int n) {
  hipLaunchKernelGGL(k_cast_c_i, dim3(ew_grid(n)), dim3(256), 0, s, out, in, n);
}
void clear_i(hipStream_t s, int *out, int n) {
  if (n > 0) hipLaunchKernelGGL(k_clear_i, dim3(ew_grid(n)), dim3(256), 0, s, out, n);
}
void copy_i(hipStream_t s, int *out, const int *in, int n) {
  if (n > 0) hipLaunchKernelGGL(k_copy_i, dim3(ew_grid(n)), dim3(256), 0, s, out, in, n);
}
void rand_i(hipStream_t s, int *out, uint64_t seed, int n) {
  if (n > 0) hipLaunchKernelGGL(k_rand_i, dim3(ew_grid(n)), dim3(256), 0, s, out, seed, n);
}

}  // namespace rdk
