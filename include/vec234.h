/*
 * rectdetect-mi355x: double-precision 2/3/4-vectors with the reference's type layout and function names
 * (reference vec234.h).  rect_t (oclrect.h) is built from vec2 / vec3, so the layout is part of the ABI.
 * Every operation is written out in the same left-to-right order as the reference evaluates it, because
 * the host post-process has to reproduce the reference's doubles bit for bit.
 */
#ifndef RD_COMPAT_VEC234_H
#define RD_COMPAT_VEC234_H
#include <math.h>
#if defined(__cplusplus)
extern "C" {
#endif

typedef struct { double a[2]; } vec2;
typedef struct { double a[3]; } vec3;
typedef struct { double a[4]; } vec4;

#define RD_VEC_OPS(N)                                                                                         \
  static inline vec##N plus##N(vec##N p, vec##N q) { vec##N r; for (int i = 0; i < N; i++) r.a[i] = p.a[i] + q.a[i]; return r; } \
  static inline vec##N minus##N(vec##N p, vec##N q) { vec##N r; for (int i = 0; i < N; i++) r.a[i] = p.a[i] - q.a[i]; return r; } \
  static inline double vdot##N(vec##N p, vec##N q) { double s = 0; for (int i = 0; i < N; i++) s += p.a[i] * q.a[i]; return s; } \
  static inline vec##N dot##N(vec##N p, double d) { vec##N r; for (int i = 0; i < N; i++) r.a[i] = p.a[i] * d; return r; }    \
  static inline double lengthSqu##N(vec##N p) { double s = 0; for (int i = 0; i < N; i++) s += p.a[i] * p.a[i]; return s; }   \
  static inline vec##N normalize##N(vec##N p) { return dot##N(p, 1.0 / (sqrt(lengthSqu##N(p)) + 1e-20)); }                \
  static inline double distanceSqu##N(vec##N p, vec##N q) { return lengthSqu##N(minus##N(p, q)); }                        \
  static inline double distance##N(vec##N p, vec##N q) { return sqrt(distanceSqu##N(p, q)); }                             \
  static inline vec##N midpoint##N(vec##N p, vec##N q) { return dot##N(plus##N(p, q), 0.5); }

RD_VEC_OPS(2)
RD_VEC_OPS(3)
RD_VEC_OPS(4)

static inline vec2 cvec2(double d0, double d1) { vec2 v; v.a[0] = d0; v.a[1] = d1; return v; }
static inline vec3 cvec3(double d0, double d1, double d2) { vec3 v; v.a[0] = d0; v.a[1] = d1; v.a[2] = d2; return v; }
static inline vec4 cvec4(double d0, double d1, double d2, double d3) { vec4 v; v.a[0] = d0; v.a[1] = d1; v.a[2] = d2; v.a[3] = d3; return v; }

#if defined(__cplusplus)
}
#endif
#endif
