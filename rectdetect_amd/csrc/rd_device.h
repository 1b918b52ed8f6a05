// rectdetect-mi355x: device-side inline helpers shared by all gfx950 kernels.
//
// Arithmetic contract: every translation unit that includes this file is
// compiled with -ffp-contract=off; sqrtf(), sqrt() and `/` are the IEEE correctly rounded forms hipcc emits by
// default (never -ffast-math, never the __f*_rn intrinsics, which map to native approximations on ROCm 7.2).
// Citations: "iu" = reference oclimgutil.cl, "rc" = oclrect.cl, "pl" = oclpolyline.cl.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define RD_WAVE 64

namespace rd {

__device__ __forceinline__ int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
__device__ __forceinline__ uint32_t clampu(uint32_t x, uint32_t lo, uint32_t hi) { return x < lo ? lo : (x > hi ? hi : x); }

// iu:41-49: reflect without repeating the border sample
__device__ __forceinline__ int mirror1(int x, int n) { return clampi(x, -x, 2 * n - 2 - x); }
__device__ __forceinline__ int mirror2(int x, int y, int iw, int ih) { return mirror1(x, iw) + mirror1(y, ih) * iw; }

// OpenCL convert_uint_rtn(float): floor, saturating at both ends
__device__ __forceinline__ uint32_t floor_u32(float f) {
  float g = floorf(f);
  if (!(g > 0.0f)) return 0u;
  if (g >= 4294967296.0f) return 0xffffffffu;
  return (uint32_t)g;
}

// iu:28-34 packed Lab: bits 0-11 L*4096, 12-21 a*1024, 22-31 b*1024
__device__ __forceinline__ uint32_t pack_lab(float L, float a, float b) {
  uint32_t r = clampu(floor_u32(b * 1024), 0u, 1023u);
  r = (r << 10) | clampu(floor_u32(a * 1024), 0u, 1023u);
  r = (r << 12) | clampu(floor_u32(L * 4096), 0u, 4095u);
  return r;
}

// iu:36-39: integer field * 2^-k + half an LSB (a multiply and an add, not fused)
__device__ __forceinline__ void unpack_lab(uint32_t p, float &L, float &a, float &b) {
  L = (float)(int)(p & 4095u) * (1.0f / 4096) + 0.5f / 4096;
  a = (float)(int)((p >> 12) & 1023u) * (1.0f / 1024) + 0.5f / 1024;
  b = (float)(int)((p >> 22) & 1023u) * (1.0f / 1024) + 0.5f / 1024;
}

// iu:65-74 Catmull-Rom style cubic through p0..p3 at fraction x (operation order is part of the result)
__device__ __forceinline__ float cubic1(float p0, float p1, float p2, float p3, float x) {
  float v = p1 - p2, w = p3 - p0;
  float u = v * 3.0f + w;
  u = u * x + (-4.0f * v + (p0 - p1 - w));
  u = u * x + (p2 - p0);
  u = u * x * 0.5f + p1;
  return u;
}

// 8-neighbour offsets in the order E, NE, N, NW, W, SW, S, SE (rc:12-13, pl:63-64); the order matters for
// "first two same-label neighbours" in the chain tracer.
__device__ __forceinline__ int nbr_dx(int i) { return (0x0 | ((i == 0 || i == 1 || i == 7) ? 1 : ((i == 3 || i == 4 || i == 5) ? -1 : 0))); }
__device__ __forceinline__ int nbr_dy(int i) { return (i >= 1 && i <= 3) ? -1 : ((i >= 5 && i <= 7) ? 1 : 0); }

// pl:870-889 per-pixel 64-bit mixing function used to break distance ties in the polyline split
__device__ __forceinline__ uint64_t rotl64(uint64_t t, int n) { n &= 63; return n ? (t << n) | (t >> (64 - n)) : t; }
__device__ __forceinline__ uint64_t mix64(uint64_t s) {
  uint64_t t = s;
  t = rotl64(t, (int)((s >> 24) & 63)); t ^= 0xf3dd0fb7820fde37ULL;
  t = rotl64(t, (int)((s >> 6) & 63));  t ^= 0xe6c6ac2c59e52811ULL;
  t = rotl64(t, (int)((s >> 18) & 63)); t ^= 0x2fc7871fff7c5b45ULL;
  t = rotl64(t, (int)((s >> 48) & 63)); t ^= 0x47c7e1f70aa4f7c5ULL;
  t = rotl64(t, (int)((s >> 0) & 63));  t ^= 0x094f02b7fb9ba895ULL;
  t = rotl64(t, (int)((s >> 12) & 63)); t ^= 0x89afda817e744570ULL;
  t = rotl64(t, (int)((s >> 36) & 63)); t ^= 0xc7277d052c7bf14bULL;
  return t;
}
__device__ __forceinline__ int pixel_rand(int p, uint64_t seed) {
  return (int)mix64(((uint64_t)(int64_t)p ^ 0xb21c2cb635b48285ULL) * 0x9b923b9cec745401ULL + (seed ^ 0x7bb93d75a79d2f15ULL) * 0x22cab58ada573a29ULL);
}

// Staging a tile: NCELLS cells, NT threads.  `where(t, a)` returns whether cell t has a source element and puts its index in
// a; `put(t, ok, v)` stores it.  All loads of a thread are issued before any value is used (the address of a cell without a
// source is clamped to element 0 and the value ignored): a block then waits for memory once, not once per step of a loop -
// these kernels run a few microseconds, a trip to HBM takes about one.
template <int NCELLS, int NT, typename T, typename Where, typename Put>
__device__ __forceinline__ void stage_cells(int tid, const T *__restrict__ in, Where where, Put put) {
  constexpr int IT = (NCELLS + NT - 1) / NT;
  T v[IT];
  bool ok[IT];
#pragma unroll
  for (int i = 0; i < IT; i++) {
    const int t = tid + i * NT;
    int a = 0;
    ok[i] = (IT * NT == NCELLS || t < NCELLS) && where(t, a);
    v[i] = in[ok[i] ? a : 0];
  }
#pragma unroll
  for (int i = 0; i < IT; i++) {
    const int t = tid + i * NT;
    if (IT * NT == NCELLS || t < NCELLS) put(t, ok[i], v[i]);
  }
}

// Element `idx` of a plane of 4-byte elements, addressed by a 32-bit BYTE offset (planes are below 2^25 elements): the access then
// takes the plane's base from scalar registers plus a 32-bit vector offset, instead of a 64-bit address computed per lane with
// sign extension, shift and 64-bit add (three vector instructions per access; a quarter of the region rounds' instruction count).
template <typename T>
__device__ __forceinline__ T &at32(T *base, unsigned idx) { return *(T *)((char *)base + (idx << 2)); }
template <typename T>
__device__ __forceinline__ const T &at32(const T *base, unsigned idx) { return *(const T *)((const char *)base + (idx << 2)); }
// the same for any element size: element `idx`, byte offset idx * sizeof(T) in 32 bits
template <typename T>
__device__ __forceinline__ T &atu(T *base, unsigned idx) { return *(T *)((char *)base + idx * (unsigned)sizeof(T)); }
template <typename T>
__device__ __forceinline__ const T &atu(const T *base, unsigned idx) { return *(const T *)((const char *)base + idx * (unsigned)sizeof(T)); }

// Frames batched per launch (small frames): frame z of a group launch works on planes that lie z * zs bytes behind the ones the launch
// was given (all planes of a frame slot are carved out of one allocation at the same offsets, rd_api.hip: slot_planes).  A kernel of the
// frame path starts with  RD_ZSHIFT(zs, out, in, ...)  on its plane pointers (null pointers - optional planes - stay null); with
// gridDim.z == 1 or zs == 0 nothing moves.  Scalar arithmetic: the offset is the same for all lanes.
template <typename T> __device__ __forceinline__ T *rd_zs_ptr(T *p, size_t off) { return p ? (T *)((char *)p + off) : p; }
#define RD_ZS1(p) (p) = rd_zs_ptr((p), rd_zoff_)
#define RD_ZS0(zs) const size_t rd_zoff_ = (size_t)blockIdx.z * (zs)
#define RD_ZS_2(zs, a) RD_ZS0(zs); RD_ZS1(a)
#define RD_ZS_3(zs, a, b) RD_ZS_2(zs, a); RD_ZS1(b)
#define RD_ZS_4(zs, a, b, c) RD_ZS_3(zs, a, b); RD_ZS1(c)
#define RD_ZS_5(zs, a, b, c, d) RD_ZS_4(zs, a, b, c); RD_ZS1(d)
#define RD_ZS_6(zs, a, b, c, d, e) RD_ZS_5(zs, a, b, c, d); RD_ZS1(e)
#define RD_ZS_7(zs, a, b, c, d, e, f) RD_ZS_6(zs, a, b, c, d, e); RD_ZS1(f)
#define RD_ZS_8(zs, a, b, c, d, e, f, g) RD_ZS_7(zs, a, b, c, d, e, f); RD_ZS1(g)
#define RD_ZS_9(zs, a, b, c, d, e, f, g, h) RD_ZS_8(zs, a, b, c, d, e, f, g); RD_ZS1(h)
#define RD_ZS_10(zs, a, b, c, d, e, f, g, h, i) RD_ZS_9(zs, a, b, c, d, e, f, g, h); RD_ZS1(i)
#define RD_ZS_PICK(_1, _2, _3, _4, _5, _6, _7, _8, _9, _10, NAME, ...) NAME
#define RD_ZSHIFT(...) RD_ZS_PICK(__VA_ARGS__, RD_ZS_10, RD_ZS_9, RD_ZS_8, RD_ZS_7, RD_ZS_6, RD_ZS_5, RD_ZS_4, RD_ZS_3, RD_ZS_2, RD_ZS_1)(__VA_ARGS__)

// Frame / XCD affinity of tile kernels.  The hardware hands the blocks of a launch to the 8 XCDs in turn (block b -> XCD b mod 8), each XCD with an L2 of its own; a
// tile kernel with a halo re-reads its neighbours' pixels, and with tile x as the fastest block coordinate every neighbour sat in another L2 (64-76 % of the
// L2 requests of such kernels missed).  These kernels are launched ONE-DIMENSIONAL over gx x gy tiles x nz frames and ask rd_block_tile() which tile they are: in
// a group launch of 8 frames block b works on frame b mod 8 - one frame per XCD, its tiles in raster order -, otherwise in plain raster order.  Speed only: nothing
// depends on where a block runs.  gdim = rd_gdim(gx, gy, nz).
// threadIdx.y of a block whose rows are waves (every 2-D block here is 64 threads wide) as a SCALAR: row numbers, row addresses and row tests that depend on
// it alone then run on the scalar unit instead of costing every lane a vector instruction
// (Only right for blocks whose rows are whole waves: 64 threads wide on a wave64 target.  gfx950 is wave64 - checked here - and every launch that uses it goes
//  through dim3(64, rows); a build with -DRD_DEBUG_LAUNCH checks blockDim.x.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "these kernels are written for gfx950 (64-wide waves: rd_ty(), the ballots and the DPP wave shifts rest on it)"
#endif
__device__ __forceinline__ int rd_ty() {
#ifdef RD_DEBUG_LAUNCH
  if (blockDim.x != 64) __builtin_trap();
#endif
  return __builtin_amdgcn_readfirstlane((int)threadIdx.y);
}
struct rd_tile { int x, y, z; };
// (12 bits each for the tile columns and rows, 7 for the frames: a launch beyond that would decode the wrong tile - the host side refuses it)
__host__ __device__ inline int rd_gdim(int gx, int gy, int nz) {
#if !defined(__HIP_DEVICE_COMPILE__)
  if (gx < 1 || gy < 1 || nz < 1 || gx >= 4096 || gy >= 4096 || nz >= 128) { fprintf(stderr, "rd_gdim: %d x %d tiles x %d frames do not fit the packed grid word (4095 x 4095 x 127)\n", gx, gy, nz); abort(); }
#endif
  return gx | (gy << 12) | (nz << 24);
}
// blocks to launch for gx x gy tiles x nz frames
__host__ __device__ inline int rd_tile_blocks(int gx, int gy, int nz) { return gx * gy * nz; }
// the block's tile (x < 0 would mean "a block of padding": none at present - other launches than 8-frame groups keep the plain raster order; giving each XCD a
// band of tile columns there measured 2 % SLOWER, one and two frames in flight and at 3840x2160 alike)
__device__ __forceinline__ rd_tile rd_block_tile(int gdim) {
  const int gx = gdim & 4095, gy = (gdim >> 12) & 4095, nz = gdim >> 24, b = (int)blockIdx.x;
  rd_tile t;
  if (nz == 8) { t.z = b & 7; const int q = b >> 3; t.y = q / gx; t.x = q - t.y * gx; }
  else { const int per = gx * gy; t.z = b / per; const int q = b - t.z * per; t.y = q / gx; t.x = q - t.y * gx; }
  return t;
}
// RD_ZSHIFT for such a kernel: the frame is rd_block_tile().z, not blockIdx.z
#define RD_ZA_1(a) RD_ZS1(a)
#define RD_ZA_2(a, b) RD_ZS1(a); RD_ZS1(b)
#define RD_ZA_3(a, b, c) RD_ZA_2(a, b); RD_ZS1(c)
#define RD_ZA_4(a, b, c, d) RD_ZA_3(a, b, c); RD_ZS1(d)
#define RD_ZA_5(a, b, c, d, e) RD_ZA_4(a, b, c, d); RD_ZS1(e)
#define RD_ZA_6(a, b, c, d, e, f) RD_ZA_5(a, b, c, d, e); RD_ZS1(f)
#define RD_ZA_7(a, b, c, d, e, f, g) RD_ZA_6(a, b, c, d, e, f); RD_ZS1(g)
#define RD_ZA_8(a, b, c, d, e, f, g, h) RD_ZA_7(a, b, c, d, e, f, g); RD_ZS1(h)
#define RD_ZA_9(a, b, c, d, e, f, g, h, i) RD_ZA_8(a, b, c, d, e, f, g, h); RD_ZS1(i)
#define RD_ZA_PICK(_1, _2, _3, _4, _5, _6, _7, _8, _9, NAME, ...) NAME
#define RD_ZSHIFTZ(z, zs, ...) const size_t rd_zoff_ = (size_t)(z) * (zs); RD_ZA_PICK(__VA_ARGS__, RD_ZA_9, RD_ZA_8, RD_ZA_7, RD_ZA_6, RD_ZA_5, RD_ZA_4, RD_ZA_3, RD_ZA_2, RD_ZA_1)(__VA_ARGS__)

// L2-coherent (device-scope) load: used to guard hot atomics so that later waves see an earlier wave's update instead of
// a stale L1 line and skip the atomic
__device__ __forceinline__ int ld_agent(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// lock-free union-find on an int label array where a root r satisfies label[r] == r and the smaller index wins
__device__ __forceinline__ int uf_find(const int *label, int a) {
  int l = label[a];
  while (l != a) { a = l; l = label[a]; }
  return a;
}
// (the walk of a union: words other threads lower meanwhile - device-scope relaxed loads of the GLOBAL plane.  A `volatile` pointer loses its address space and
//  made every step a flat load with system-scope bits.)
__device__ __forceinline__ int uf_find_volatile(const int *label, int a) {
  int l = ld_agent(label + a);
  while (l != a) { a = l; l = ld_agent(label + a); }
  return a;
}
__device__ __forceinline__ void uf_union(int *label, int a, int b) {
  for (;;) {
    a = uf_find_volatile(label, a);
    b = uf_find_volatile(label, b);
    if (a == b) return;
    if (a < b) { int t = a; a = b; b = t; }   // a > b: hang a under b
    int old = atomicMin(&label[a], b);
    if (old == a) return;                      // a was still a root: linked
    a = old;                                   // somebody re-parented a meanwhile: retry from there
  }
}

}  // namespace rd
