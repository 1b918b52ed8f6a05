/*
 * TEST INFRASTRUCTURE ONLY (oracle/). Never linked into the product library.
 *
 * OpenCL C builtin functions for the reference's kernels when those kernels are
 * compiled to x86-64 objects (clang -x cl, see oracle/Makefile).  Every unresolved
 * symbol of oclimgutil.cl / oclpolyline.cl / oclrect.cl is an Itanium-mangled OpenCL
 * builtin (SURVEY.md Appendix B lists all 29); they are defined here with asm labels.
 * Must be compiled with the same clang as the kernels so ext_vector_type ABIs agree.
 *
 * These definitions ARE the canonical arithmetic of the oracle ("serial CPU device"):
 *   - atomics: plain read-modify-write returning the old value (serial execution)
 *   - convert_*_rtn: floor; convert_*_rte: round-half-even; plain convert: C cast
 *   - sqrt: correctly rounded; rsqrt(x) = 1.0f / sqrtf(x); hypot via double
 *   - distance: sqrtf of the left-to-right sum of squares (no FMA: -ffp-contract=off)
 * The HIP kernels and the C restatement (oracle/rd_oracle.c) use the same definitions.
 *
 * RDCL_VARIANT (build-time bit mask, default 0 = the definitions above): OTHER choices an OpenCL device may legally make for the builtins whose
 * accuracy the standard leaves loose, so that it can be MEASURED how much of "the reference's output" hangs on ours
 * (tools/make_golden_builtins.py -> tests/golden/builtin_sensitivity.npz):
 *   1  rsqrt correctly rounded in one step ((float)(1 / sqrt((double)x))) instead of two correctly rounded operations
 *   2  hypot evaluated in float (sqrtf(a * a + b * b)) instead of through double
 *   4  distance through double (one rounding) instead of float operations left to right
 * (FMA contraction - OpenCL C's default, H11 - is a compiler flag of the kernel objects, not a builtin: oracle/Makefile, variant "fma".)
 */
#ifndef RDCL_VARIANT
#define RDCL_VARIANT 0
#endif
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

typedef int int2 __attribute__((ext_vector_type(2)));
typedef int int3 __attribute__((ext_vector_type(3)));
typedef short short2 __attribute__((ext_vector_type(2)));
typedef float float2 __attribute__((ext_vector_type(2)));
typedef float float3 __attribute__((ext_vector_type(3)));

/* Work-item id of the currently executing work-item; written by the dispatcher
 * (rdcl_device.c) through dlsym("rdcl_gid") before every kernel call. */
size_t rdcl_gid[3];

size_t b_get_global_id(unsigned d) __asm__("_Z13get_global_idj");
size_t b_get_global_id(unsigned d) { return d < 3 ? rdcl_gid[d] : 0; }

int b_atomic_add_i(volatile int *p, int v) __asm__("_Z10atomic_addPU8CLglobalVii");
int b_atomic_add_i(volatile int *p, int v) { int o = *p; *p = o + v; return o; }

unsigned b_atomic_add_u(volatile unsigned *p, unsigned v) __asm__("_Z10atomic_addPU8CLglobalVjj");
unsigned b_atomic_add_u(volatile unsigned *p, unsigned v) { unsigned o = *p; *p = o + v; return o; }

int b_atomic_inc(volatile int *p) __asm__("_Z10atomic_incPU8CLglobalVi");
int b_atomic_inc(volatile int *p) { int o = *p; *p = o + 1; return o; }

int b_atomic_max(volatile int *p, int v) __asm__("_Z10atomic_maxPU8CLglobalVii");
int b_atomic_max(volatile int *p, int v) { int o = *p; if (v > o) *p = v; return o; }

/* "Concurrent" launches (rdcl_set_order group_order 5, rdcl_device.c): every work-item sees memory as it was when the launch
 * began - a device gives no guarantee that one work-item sees another's update within a launch - and the atomic minima of all
 * work-items take effect together when it ends.  rdcl_defer_atomic_min(1) starts logging, (0) applies the log. */
static int defer_min_on = 0;
static struct { volatile int *p; int v; } *defer_log = 0;
static size_t defer_n = 0, defer_cap = 0;
void rdcl_defer_atomic_min(int on) {
  if (!on) {
    for (size_t i = 0; i < defer_n; i++) if (defer_log[i].v < *defer_log[i].p) *defer_log[i].p = defer_log[i].v;
    defer_n = 0;
  }
  defer_min_on = on;
}

int b_atomic_min(volatile int *p, int v) __asm__("_Z10atomic_minPU8CLglobalVii");
int b_atomic_min(volatile int *p, int v) {
  int o = *p;
  if (defer_min_on) {
    if (defer_n == defer_cap) { defer_cap = defer_cap ? defer_cap * 2 : (1u << 20); defer_log = realloc(defer_log, defer_cap * sizeof(*defer_log)); }
    defer_log[defer_n].p = p; defer_log[defer_n].v = v; defer_n++;
  } else if (v < o) *p = v;
  return o;
}

int b_atomic_cmpxchg(volatile int *p, int cmp, int v) __asm__("_Z14atomic_cmpxchgPU8CLglobalViii");
int b_atomic_cmpxchg(volatile int *p, int cmp, int v) { int o = *p; if (o == cmp) *p = v; return o; }

int b_clamp_i(int x, int lo, int hi) __asm__("_Z5clampiii");
int b_clamp_i(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

unsigned b_clamp_u(unsigned x, unsigned lo, unsigned hi) __asm__("_Z5clampjjj");
unsigned b_clamp_u(unsigned x, unsigned lo, unsigned hi) { return x < lo ? lo : (x > hi ? hi : x); }

int2 b_clamp_i2(int2 x, int2 lo, int2 hi) __asm__("_Z5clampDv2_iS_S_");
int2 b_clamp_i2(int2 x, int2 lo, int2 hi) {
  int2 r;
  r.x = x.x < lo.x ? lo.x : (x.x > hi.x ? hi.x : x.x);
  r.y = x.y < lo.y ? lo.y : (x.y > hi.y ? hi.y : x.y);
  return r;
}

unsigned b_abs_diff(int a, int b) __asm__("_Z8abs_diffii");
unsigned b_abs_diff(int a, int b) { return a > b ? (unsigned)a - (unsigned)b : (unsigned)b - (unsigned)a; }

int b_convert_int_rtn(float f) __asm__("_Z15convert_int_rtnf");
int b_convert_int_rtn(float f) { return (int)floorf(f); }

unsigned b_convert_uint_rtn_f(float f) __asm__("_Z16convert_uint_rtnf");
unsigned b_convert_uint_rtn_f(float f) {
  float g = floorf(f);
  if (!(g > 0.0f)) return 0u;
  if (g >= 4294967296.0f) return 0xffffffffu;
  return (unsigned)g;
}

unsigned b_convert_uint_rtn_i(int i) __asm__("_Z16convert_uint_rtni");
unsigned b_convert_uint_rtn_i(int i) { return (unsigned)i; }

float3 b_convert_float3(int3 v) __asm__("_Z14convert_float3Dv3_i");
float3 b_convert_float3(int3 v) { float3 r; r.x = (float)v.x; r.y = (float)v.y; r.z = (float)v.z; return r; }

float2 b_convert_float2_s(short2 v) __asm__("_Z14convert_float2Dv2_s");
float2 b_convert_float2_s(short2 v) { float2 r; r.x = (float)v.x; r.y = (float)v.y; return r; }

short2 b_convert_short2(float2 v) __asm__("_Z14convert_short2Dv2_f");
short2 b_convert_short2(float2 v) { short2 r; r.x = (short)v.x; r.y = (short)v.y; return r; }

int2 b_convert_int2_s(short2 v) __asm__("_Z12convert_int2Dv2_s");
int2 b_convert_int2_s(short2 v) { int2 r; r.x = (int)v.x; r.y = (int)v.y; return r; }

int2 b_convert_int2_rte(float2 v) __asm__("_Z16convert_int2_rteDv2_f");
int2 b_convert_int2_rte(float2 v) { int2 r; r.x = (int)rintf(v.x); r.y = (int)rintf(v.y); return r; }

long b_convert_long_rte(float f) __asm__("_Z16convert_long_rtef");
long b_convert_long_rte(float f) { return (long)rintf(f); }

float b_max_f(float a, float b) __asm__("_Z3maxff");
float b_max_f(float a, float b) { return fmaxf(a, b); }

float3 b_max_f3(float3 a, float3 b) __asm__("_Z3maxDv3_fS_");
float3 b_max_f3(float3 a, float3 b) {
  float3 r; r.x = fmaxf(a.x, b.x); r.y = fmaxf(a.y, b.y); r.z = fmaxf(a.z, b.z); return r;
}

float b_sqrt(float x) __asm__("_Z4sqrtf");
float b_sqrt(float x) { return sqrtf(x); }

float b_rsqrt(float x) __asm__("_Z5rsqrtf");
#if RDCL_VARIANT & 1
float b_rsqrt(float x) { return (float)(1.0 / sqrt((double)x)); }
#else
float b_rsqrt(float x) { return 1.0f / sqrtf(x); }
#endif

float b_hypot(float a, float b) __asm__("_Z5hypotff");
#if RDCL_VARIANT & 2
float b_hypot(float a, float b) { return sqrtf(a * a + b * b); }
#else
float b_hypot(float a, float b) { return (float)sqrt((double)a * (double)a + (double)b * (double)b); }
#endif

float b_fabs(float x) __asm__("_Z4fabsf");
float b_fabs(float x) { return fabsf(x); }

float b_round(float x) __asm__("_Z5roundf");
float b_round(float x) { return roundf(x); }

float b_distance2(float2 a, float2 b) __asm__("_Z8distanceDv2_fS_");
float b_distance2(float2 a, float2 b) {
  float dx = a.x - b.x, dy = a.y - b.y;
#if RDCL_VARIANT & 4
  return (float)sqrt((double)dx * dx + (double)dy * dy);
#else
  return sqrtf(dx * dx + dy * dy);
#endif
}

float b_distance3(float3 a, float3 b) __asm__("_Z8distanceDv3_fS_");
float b_distance3(float3 a, float3 b) {
  float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
#if RDCL_VARIANT & 4
  return (float)sqrt((double)dx * dx + (double)dy * dy + (double)dz * dz);
#else
  return sqrtf(dx * dx + dy * dy + dz * dz);
#endif
}
