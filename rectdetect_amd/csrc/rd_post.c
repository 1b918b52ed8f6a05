/*
 * rectdetect-mi355x: host post-process - line segments + segment/boundary votes -> rectangles.
 *
 * Restates the behaviour of the reference's executeCPUTask and its helpers (oclrect.c:385-1226, "rh") in double
 * precision with the same operation order, so that the returned rect_t lists are bit-identical to the reference's for
 * identical inputs (tests compare them against the reference build).  What differs is the INPUT format: instead of the
 * 4N-byte boundary-label plane and the 16N-byte voting table read back over PCIe (rh:371-376), the device hands over,
 * per segment, the 15 probes of rh:1066-1098 already resolved: {boundary id, table slot {owner, 4 box values}}.
 *
 * Build with -ffp-contract=off (no FMA): see rd_device.h for the arithmetic contract.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define CL_TARGET_OPENCL_VERSION 120
#include <CL/cl.h>
#include "helper.h"
#include "vec234.h"
#include "oclpolyline.h"
#include "oclrect.h"
#include "rd_post.h"

/* ------------------------------------------------------------------ small containers */

typedef struct { vec2 e0, e1; } seg2;
typedef struct { seg2 *v; int n, cap; } seglist;
typedef struct { vec2 *v; int n, cap; } ptlist;
typedef struct { int *v; int n, cap; } intlist;

static void sl_push(seglist *l, seg2 s) {
  if (l->n == l->cap) { l->cap = l->cap ? l->cap * 2 : 16; l->v = (seg2 *)realloc(l->v, sizeof(seg2) * (size_t)l->cap); }
  l->v[l->n++] = s;
}
static void sl_remove(seglist *l, int i) { memmove(l->v + i, l->v + i + 1, sizeof(seg2) * (size_t)(l->n - i - 1)); l->n--; }
static void pl_push(ptlist *l, vec2 p) {
  if (l->n == l->cap) { l->cap = l->cap ? l->cap * 2 : 16; l->v = (vec2 *)realloc(l->v, sizeof(vec2) * (size_t)l->cap); }
  l->v[l->n++] = p;
}
static void il_push(intlist *l, int x) {
  if (l->n == l->cap) { l->cap = l->cap ? l->cap * 2 : 16; l->v = (int *)realloc(l->v, sizeof(int) * (size_t)l->cap); }
  l->v[l->n++] = x;
}

/* ------------------------------------------------------------------ 2-D geometry (rh:389-425) */

static inline double sq(double x) { return x * x; }
static float seg_sqlen(const seg2 *s) { return (float)distanceSqu2(s->e0, s->e1); }   /* rh:390: narrowed to float */

/* foot of the perpendicular from p on the LINE through v, w (rh:400-406) */
static vec2 foot_on_line(vec2 v, vec2 w, vec2 p) {
  double l2 = distanceSqu2(v, w);
  if (l2 == 0.0) return v;
  double t = ((p.a[0] - v.a[0]) * (w.a[0] - v.a[0]) + (p.a[1] - v.a[1]) * (w.a[1] - v.a[1])) / l2;
  return cvec2(v.a[0] + t * (w.a[0] - v.a[0]), v.a[1] + t * (w.a[1] - v.a[1]));
}

/* closest point of the SEGMENT v-w to p (rh:408-416) */
static vec2 closest_on_seg(vec2 v, vec2 w, vec2 p) {
  double l2 = distanceSqu2(v, w);
  if (l2 == 0.0) return v;
  double t = ((p.a[0] - v.a[0]) * (w.a[0] - v.a[0]) + (p.a[1] - v.a[1]) * (w.a[1] - v.a[1])) / l2;
  if (t < 0) return v;
  if (t > 1.0) return w;
  return cvec2(v.a[0] + t * (w.a[0] - v.a[0]), v.a[1] + t * (w.a[1] - v.a[1]));
}

/* intersection of the lines through u and v; NaN when (nearly) parallel (rh:418-425) */
static vec2 line_intersection(seg2 u, seg2 v) {
  double d = (v.e1.a[0] - v.e0.a[0]) * (u.e1.a[1] - u.e0.a[1]) - (v.e1.a[1] - v.e0.a[1]) * (u.e1.a[0] - u.e0.a[0]);
  if (fabs(d) < 1e-4) return cvec2(NAN, NAN);
  double n = (v.e0.a[1] - u.e0.a[1]) * (u.e1.a[0] - u.e0.a[0]) - (v.e0.a[0] - u.e0.a[0]) * (u.e1.a[1] - u.e0.a[1]);
  double q = n / d;
  return cvec2(v.e0.a[0] + q * (v.e1.a[0] - v.e0.a[0]), v.e0.a[1] + q * (v.e1.a[1] - v.e0.a[1]));
}

static vec3 cross3v(vec3 v, vec3 w) {
  return cvec3(v.a[1] * w.a[2] - v.a[2] * w.a[1], v.a[2] * w.a[0] - v.a[0] * w.a[2], v.a[0] * w.a[1] - v.a[1] * w.a[0]);
}

/* ------------------------------------------------------------------ pose estimation (rh:427-634) */
/* Unknowns: the depths of the four corner rays.  The residual asks for a planar rectangle with unit sides in one of two
 * pairings (mode).  Minimised by a diagonally preconditioned non-linear CG with numerical derivatives. */

typedef struct { const vec3 *ray; int mode; } pose_arg;

#define POSE_EPS (1e-6)

static double pose_residual(vec4 v, const pose_arg *arg) {
  const vec3 *ray = arg->ray;
  const int mode = arg->mode;
  vec3 q[4];
  for (int i = 0; i < 4; i++) q[i] = dot3(ray[i], v.a[i]);

  double score = 0;
  const double l01 = distanceSqu3(q[0], q[1]), l12 = distanceSqu3(q[1], q[2]), l23 = distanceSqu3(q[2], q[3]);
  const double l03 = distanceSqu3(q[0], q[3]), l02 = distanceSqu3(q[0], q[2]), l13 = distanceSqu3(q[1], q[3]);

  score += sq((mode ? l23 : l03) - 1);
  score += sq((mode ? l01 : l12) - 1);
  const double comp = 1.0 / (mode ? l12 : l01);

  score += lengthSqu3(plus3(minus3(mode ? q[0] : q[2], q[1]), minus3(mode ? q[2] : q[0], q[3])));
  score += comp * lengthSqu3(plus3(minus3(q[1], mode ? q[2] : q[0]), minus3(q[3], mode ? q[0] : q[2])));

  score += sq(l01 + l12 - l02);
  score += sq(l03 + l23 - l02);
  score += sq(l01 + l03 - l13);
  score += sq(l12 + l23 - l13);

  const vec3 n013 = cross3v(minus3(q[1], q[0]), minus3(q[3], q[0]));
  score += comp * sq(vdot3(n013, q[2]) - vdot3(n013, q[0])) / vdot3(n013, n013);
  const vec3 n102 = cross3v(minus3(q[0], q[1]), minus3(q[2], q[1]));
  score += comp * sq(vdot3(n102, q[3]) - vdot3(n102, q[1])) / vdot3(n102, n102);
  return score;
}

/* value, first and second directional derivative along dir (rh:479-490) */
static vec3 directional(vec4 v, vec4 dir, const pose_arg *arg) {
  const double h = POSE_EPS;
  const double f0 = pose_residual(v, arg);
  const double fp = pose_residual(plus4(v, dot4(dir, h)), arg);
  const double fm = pose_residual(plus4(v, dot4(dir, -h)), arg);
  return cvec3(f0, (fp - fm) * (1.0 / (2 * h)), (fp + fm - 2 * f0) * (1.0 / (h * h)));
}

/* per-coordinate first and second derivatives (rh:492-512) */
static void coordinate_derivs(vec4 v, const pose_arg *arg, vec4 *g, vec4 *g2) {
  const double fx = pose_residual(v, arg);
  for (int i = 0; i < 4; i++) {
    vec4 d = cvec4(0, 0, 0, 0);
    d.a[i] = POSE_EPS;
    const double fm = pose_residual(minus4(v, d), arg);
    const double fp = pose_residual(plus4(v, d), arg);
    g->a[i] = (fp - fm) / (2 * POSE_EPS);
    g2->a[i] = (fm - 2 * fx + fp) / (POSE_EPS * POSE_EPS);
  }
}

/* Newton steps along dir with step halving (rh:514-536) */
static vec4 line_search(vec4 iv, vec4 dir, int iters, const pose_arg *arg) {
  dir = normalize4(dir);
  double scale = 1.0;
  for (int i = 0; i < iters; i++) {
    vec3 gd = directional(iv, dir, arg);
    const double ep = gd.a[0];
    if (gd.a[2] * gd.a[2] < 1e-10) gd.a[2] = 1;
    const double delta = fabs(gd.a[1] / gd.a[2]);
    if (delta < 1e-10) return iv;
    const vec4 v = plus4(iv, dot4(dir, delta * scale));
    const double e1 = pose_residual(v, arg);
    if (ep < e1) { scale *= 0.5; continue; }
    iv = v;
  }
  return iv;
}

/* r / m per coordinate when every m is positive, else r (rh:538-555) */
static vec4 precondition(vec4 m, vec4 r) {
  for (int i = 0; i < 4; i++) if (m.a[i] <= 0) return r;
  vec4 a;
  for (int i = 0; i < 4; i++) { a.a[i] = 1.0 / m.a[i]; a.a[i] *= r.a[i]; }
  return a;
}

/* rh:557-588 */
static vec4 conjugate_gradient(vec4 x, int loops, int ls_iters, const pose_arg *arg) {
  vec4 g, g2;
  coordinate_derivs(x, arg, &g, &g2);
  vec4 r = dot4(g, -1), m = g2;
  vec4 s = precondition(m, r), d = s;
  double deltanew = vdot4(r, d);
  int k = 0;
  for (int i = 0; i < loops; i++) {
    x = line_search(x, d, ls_iters, arg);
    coordinate_derivs(x, arg, &g, &g2);
    r = dot4(g, -1); m = g2;
    const double deltaold = deltanew;
    const double deltamid = vdot4(r, s);
    s = precondition(m, r);
    deltanew = vdot4(r, s);
    const double beta = (deltanew - deltamid) / deltaold;
    if (k == 10 || beta <= 0 || deltaold == 0) { d = s; k = 0; }
    else d = plus4(s, dot4(d, beta));
    k++;
  }
  return x;
}

/* rh:590-634: als = the four sides in angular order, centre = their length-weighted centroid */
static rect_t estimate_pose(const seg2 *als, vec2 centre, int iw, int ih, double tanAOV) {
  vec3 p[4];
  int tl = 0;
  double min = 1e+100;
  for (int i = 0; i < 4; i++) {
    vec2 v = normalize2(minus2(als[i].e1, als[i].e0));
    v = cvec2(-v.a[1], v.a[0]);
    if (vdot2(minus2(als[i].e0, centre), v) < 0) v = dot2(v, -1);
    if (v.a[1] < min) { min = v.a[1]; tl = i; }
  }
  for (int i = 0; i < 4; i++) {
    const seg2 *s = &als[(i + tl) & 3];
    p[i] = normalize3(cvec3((s->e0.a[0] - (iw / 2)), (-(s->e0.a[1] - ih / 2)), iw / 2 / tanAOV));
  }
  const double d01 = 1.0 / distance3(p[0], p[1]), d23 = 1.0 / distance3(p[2], p[3]);
  pose_arg a0 = { p, 1 };
  const vec4 x0 = conjugate_gradient(cvec4(d01, d01, d23, d23), 12, 10, &a0);
  const double v0 = pose_residual(x0, &a0);
  const double d12 = 1.0 / distance3(p[1], p[2]), d03 = 1.0 / distance3(p[0], p[3]);
  pose_arg a1 = { p, 0 };
  const vec4 x1 = conjugate_gradient(cvec4(d03, d12, d12, d03), 12, 10, &a1);
  const double v1 = pose_residual(x1, &a1);

  rect_t ret;
  memset(&ret, 0, sizeof(ret));
  ret.value = v0 < v1 ? v0 : v1;
  vec4 x = v0 < v1 ? x0 : x1;
  if (x.a[0] < 0) x = dot4(x, -1);
  for (int i = 0; i < 4; i++) {
    ret.c3[i] = dot3(p[i], x.a[i]);
    ret.c2[i] = als[(i + tl) & 3].e0;
  }
  return ret;
}

/* rh:636-656 */
static int looks_like_a_screen(const rect_t *r) {
  if (r->value > 0.05) return 0;
  for (int i = 0; i < 4; i++) if (r->c3[i].a[2] < 0) return 0;
  const double asp = distance3(r->c3[0], r->c3[1]) / distance3(r->c3[1], r->c3[2]);
  if (asp < 1.0 / 12 || 12 < asp) return 0;
  double maxs = 0, mins = 1e+100;
  for (int i = 0; i < 4; i++) {
    const vec2 a = r->c2[i], b = r->c2[(i + 1) % 4], c = r->c2[(i + 2) % 4], d = r->c2[(i + 3) % 4];
    const double s0 = distanceSqu2(c, closest_on_seg(a, b, c));
    const double s1 = distanceSqu2(d, closest_on_seg(a, b, d));
    maxs = fmax(maxs, fmax(s0, s1));
    mins = fmin(mins, fmax(s0, s1));
  }
  return maxs / mins > 100 ? 0 : 1;
}

/* ------------------------------------------------------------------ convex hull (rh:658-734) */

static void hull_side(ptlist *hull, const ptlist *s, vec2 left, vec2 right) {
  int far = -1;
  double d = 0;
  for (int i = 0; i < s->n; i++) {
    const double e = distanceSqu2(foot_on_line(left, right, s->v[i]), s->v[i]);
    if (far < 0 || e > d) { far = i; d = e; }
  }
  if (d < 0.01 || far < 0) return;
  const vec2 pf = s->v[far];
  const vec2 nr = cvec2(pf.a[1] - right.a[1], right.a[0] - pf.a[0]);
  const vec2 nl = cvec2(left.a[1] - pf.a[1], pf.a[0] - left.a[0]);
  ptlist sr = { 0 }, slft = { 0 };
  for (int i = 0; i < s->n; i++) {
    if (i == far) continue;
    if (vdot2(minus2(s->v[i], pf), nr) > 0) pl_push(&sr, s->v[i]);
    if (vdot2(minus2(s->v[i], pf), nl) > 0) pl_push(&slft, s->v[i]);
  }
  hull_side(hull, &sr, pf, right);
  pl_push(hull, pf);
  hull_side(hull, &slft, left, pf);
  free(sr.v); free(slft.v);
}

static ptlist quick_hull(const ptlist *s) {
  ptlist hull = { 0 };
  if (s->n == 0) return hull;
  vec2 right = s->v[0], left = s->v[0];
  for (int i = 0; i < s->n; i++) {
    if (s->v[i].a[0] > right.a[0]) right = s->v[i];
    if (s->v[i].a[0] < left.a[0]) left = s->v[i];
  }
  const vec2 up = cvec2(left.a[1] - right.a[1], right.a[0] - left.a[0]);
  ptlist top = { 0 }, bot = { 0 };
  for (int i = 0; i < s->n; i++) {
    const vec2 p = s->v[i];
    if (p.a[0] == left.a[0] && p.a[1] == left.a[1]) continue;
    if (p.a[0] == right.a[0] && p.a[1] == right.a[1]) continue;
    if (vdot2(minus2(p, left), up) > 0) pl_push(&top, p); else pl_push(&bot, p);
  }
  pl_push(&hull, right);
  hull_side(&hull, &top, left, right);
  pl_push(&hull, left);
  hull_side(&hull, &bot, right, left);
  free(top.v); free(bot.v);
  return hull;
}

/* ------------------------------------------------------------------ Cohen-Sutherland clip (rh:744-802) */

static int outcode(double x, double y, double xmin, double ymin, double xmax, double ymax) {
  int c = 0;
  if (x < xmin) c |= 1;
  if (x > xmax) c |= 2;
  if (y < ymin) c |= 4;
  if (y > ymax) c |= 8;
  return c;
}

static int clip_to_box(double *x0, double *y0, double *x1, double *y1, double xmin, double ymin, double xmax, double ymax) {
  int c0 = outcode(*x0, *y0, xmin, ymin, xmax, ymax), c1 = outcode(*x1, *y1, xmin, ymin, xmax, ymax);
  for (;;) {
    if ((c0 | c1) == 0) return 1;
    if ((c0 & c1) != 0) return 0;
    double x = 0, y = 0;
    const int co = c0 != 0 ? c0 : c1;
    if (co & 8) { x = *x0 + (*x1 - *x0) * (ymax - *y0) / (*y1 - *y0); y = ymax; }
    else if (co & 4) { x = *x0 + (*x1 - *x0) * (ymin - *y0) / (*y1 - *y0); y = ymin; }
    else if (co & 2) { y = *y0 + (*y1 - *y0) * (xmax - *x0) / (*x1 - *x0); x = xmax; }
    else if (co & 1) { y = *y0 + (*y1 - *y0) * (xmin - *x0) / (*x1 - *x0); x = xmin; }
    if (co == c0) { *x0 = x; *y0 = y; c0 = outcode(*x0, *y0, xmin, ymin, xmax, ymax); }
    else { *x1 = x; *y1 = y; c1 = outcode(*x1, *y1, xmin, ymin, xmax, ymax); }
  }
}

/* ------------------------------------------------------------------ the candidate funnel (rh:806-1045) */

/* stable ascending sort by the float squared length (the reference uses glibc qsort, a stable merge sort) */
static void sort_by_length(seglist *l) {
  for (int i = 1; i < l->n; i++) {
    seg2 k = l->v[i];
    const float kl = seg_sqlen(&k);
    int j = i - 1;
    while (j >= 0 && seg_sqlen(&l->v[j]) > kl) { l->v[j + 1] = l->v[j]; j--; }
    l->v[j + 1] = k;
  }
}

static double outward_angle(const seg2 *s, vec2 c) {
  vec2 v = minus2(s->e0, s->e1);
  v = cvec2(v.a[1], -v.a[0]);
  if (vdot2(v, minus2(s->e0, c)) < 0) v = dot2(v, -1);
  return atan2(v.a[0], v.a[1]);
}

/* rh:821-852: stable ascending sort by the direction of the outward normal */
static void sort_by_angle(seglist *l, vec2 c) {
  for (int i = 1; i < l->n; i++) {
    seg2 k = l->v[i];
    const double ka = outward_angle(&k, c);
    int j = i - 1;
    while (j >= 0 && outward_angle(&l->v[j], c) > ka) { l->v[j + 1] = l->v[j]; j--; }
    l->v[j + 1] = k;
  }
}

/* rh:864-877 */
static vec2 weighted_centre(const seglist *l) {
  vec2 g = cvec2(0, 0);
  double sum = 0;
  for (int i = 0; i < l->n; i++) {
    const double len = distance2(l->v[i].e0, l->v[i].e1);
    g = plus2(g, dot2(plus2(l->v[i].e0, l->v[i].e1), len));
    sum += len;
  }
  return dot2(g, 0.5 / sum);
}

/* rh:879-884 */
static double total_length(const seglist *l) {
  double r = 0;
  for (int i = 0; i < l->n; i++) r += sqrt(seg_sqlen(&l->v[i]));
  return r;
}

/* rh:886-895 */
static int nearly_triangle(const seglist *l, double ratio) {
  for (int i = 0; i < l->n; i++) {
    const seg2 a = l->v[i], b = l->v[(i + 1) % l->n];
    const double d0 = distanceSqu2(a.e1, foot_on_line(a.e0, b.e1, a.e1));
    const double d1 = distanceSqu2(a.e0, b.e1);
    if (d0 / d1 < ratio) return 1;
  }
  return 0;
}

/* rh:897-922 */
static int is_convex(const seglist *l) {
  const int n = l->n;
  int sign = 0;
  for (int i = 0; i < n; i++) {
    const seg2 a = l->v[i], b = l->v[(i + 1) % n];
    const double ax = a.e1.a[0] - a.e0.a[0], ay = a.e1.a[1] - a.e0.a[1];
    const double bx = b.e1.a[0] - b.e0.a[0], by = b.e1.a[1] - b.e0.a[1];
    const int sg = ax * by - ay * bx > 0;
    if (i == 0) sign = sg;
    else if (sg != sign) return 0;
  }
  return 1;
}

/* rh:926-943 */
static void drop_short(seglist *l, float ratio) {
  if (l->n <= 4) return;
  sort_by_length(l);
  const float longest = seg_sqlen(&l->v[l->n - 1]);
  while (l->n > 4) {
    const float shortest = seg_sqlen(&l->v[0]);
    if (shortest / longest > ratio * ratio) break;
    sl_remove(l, 0);
  }
}

/* rh:945-992: keep, for every hull edge, the longest segment lying on it.  Consumes *in. */
static seglist keep_outer(seglist *in) {
  ptlist pts = { 0 };
  for (int i = 0; i < in->n; i++) { pl_push(&pts, in->v[i].e0); pl_push(&pts, in->v[i].e1); }
  ptlist q = quick_hull(&pts);
  seglist out = { 0 };
  for (int i = 0; i < q.n; i++) {
    const vec2 q0 = q.v[i], q1 = q.v[(i + 1) % q.n];
    const vec2 m = midpoint2(q0, q1), nq = normalize2(minus2(q0, q1));
    int added = -1;
    sort_by_length(in);
    for (int j = in->n - 1; j >= 0; j--) {
      const seg2 e = in->v[j];
      if (distanceSqu2(m, closest_on_seg(e.e0, e.e1, m)) < 1) { sl_push(&out, e); added = j; break; }
      if (fabs(vdot2(nq, normalize2(minus2(e.e0, e.e1)))) > 0.95 &&
          distanceSqu2(m, closest_on_seg(e.e0, e.e1, m)) / distanceSqu2(q0, q1) < 0.01) { sl_push(&out, e); added = j; break; }
    }
    if (added != -1) sl_remove(in, added);
  }
  free(q.v); free(pts.v); free(in->v);
  in->v = NULL; in->n = in->cap = 0;
  return out;
}

/* rh:994-1009.  Consumes *in when it has more than k elements. */
static seglist keep_longest(seglist *in, int k) {
  if (in->n <= k) { seglist r = *in; in->v = NULL; in->n = in->cap = 0; return r; }
  sort_by_length(in);
  seglist out = { 0 };
  for (int j = in->n - 1; j >= 0; j--) { sl_push(&out, in->v[j]); if (out.n == k) break; }
  free(in->v); in->v = NULL; in->n = in->cap = 0;
  return out;
}

/* rh:1011-1045: consecutive sides -> corners -> sides between corners; returns 0 if two sides are parallel */
static int sides_to_corners(seglist *l) {
  const int n = l->n;
  vec2 *c = (vec2 *)malloc(sizeof(vec2) * (size_t)(n ? n : 1));
  for (int i = 0; i < n; i++) {
    c[i] = line_intersection(l->v[i], l->v[(i + 1) % n]);
    if (isnan(c[i].a[0])) { free(c); return 0; }
  }
  for (int i = 0; i < n; i++) { l->v[i].e0 = c[i]; l->v[i].e1 = c[(i + 1) % n]; }
  free(c);
  return 1;
}

/* rh:1134-1160 / rh:1190-1216: returns 1 and fills *out when the candidate set yields a rectangle.  Consumes *als. */
static int funnel(seglist *als, int iw, int ih, double tanAOV, uint32_t status, rect_t *out) {
  drop_short(als, 0.05f);
  seglist outer = keep_outer(als);
  const double len0 = total_length(&outer);
  seglist four = keep_longest(&outer, 4);
  if (four.n > 0) sort_by_angle(&four, weighted_centre(&four));
  int ok = four.n > 0 ? sides_to_corners(&four) : 1;
  /* (with no segment left the reference still goes on: findCorners returns an empty list, the size test rejects it) */
  double len1 = ok ? total_length(&four) : 0;
  if (!ok || nearly_triangle(&four, 0.001) || four.n < 4 || len1 / len0 > 2 || !is_convex(&four)) { free(four.v); return 0; }
  *out = estimate_pose(four.v, weighted_centre(&four), iw, ih, tanAOV);
  out->status = status;
  if (looks_like_a_screen(out)) out->status |= 1;
  free(four.v);
  return 1;
}

/* ------------------------------------------------------------------ rh:1049-1226 */

void *rd_post_run(const void *segs, int max_records, const int *probes, int iw, int ih, double tanAOV) {
  const linesegment_t *ls = (const linesegment_t *)segs;
  int n = ((const int *)segs)[0];
  if (n > max_records - 1) n = max_records - 1;
  if (n < 0) n = 0;
  const unsigned nentry = (unsigned)(iw * ih * 4 / 5);

  rect_t *ret = (rect_t *)calloc(16, sizeof(rect_t));
  int nret = 1, capret = 16;
#define PUSH_RECT(r) do { if (nret == capret) { capret *= 2; ret = (rect_t *)realloc(ret, sizeof(rect_t) * (size_t)capret); } ret[nret++] = (r); } while (0)

  /* pass 1: segments grouped by the boundary component they run along */
  ArrayMap *groups = initArrayMap();
  for (int i = 1; i <= n; i++) {
    if (ls[i].polyid == 0) continue;
    for (int k = 0; k < 15; k++) {
      const int segid = probes[(size_t)(i * 15 + k) * 6];
      if (segid <= 0) continue;
      intlist *set = (intlist *)ArrayMap_get(groups, (uint64_t)segid);
      if (!set) { set = (intlist *)calloc(1, sizeof(intlist)); ArrayMap_put(groups, (uint64_t)segid, set); }
      int j;
      for (j = 0; j < set->n; j++) if (set->v[j] == i) break;
      if (j == set->n) il_push(set, i);
    }
  }

  uint64_t *keys = ArrayMap_keyArray(groups);
  const int ngroups = ArrayMap_size(groups);
  for (int gi = 0; gi < ngroups; gi++) {
    const int segid = (int)keys[gi];
    intlist *set = (intlist *)ArrayMap_get(groups, (uint64_t)segid);
    if (set->n < 4) continue;
    seglist als = { 0 };
    for (int j = 0; j < set->n; j++) {
      const int lsid = set->v[j];
      /* the voting-table slot of (lsid, segid): any probe of lsid that hit segid carries it */
      const int *e = NULL;
      for (int k = 0; k < 15; k++) { const int *pr = probes + (size_t)(lsid * 15 + k) * 6; if (pr[0] == segid) { e = pr + 1; break; } }
      (void)nentry;
      const seg2 whole = { cvec2(ls[lsid].x0, ls[lsid].y0), cvec2(ls[lsid].x1, ls[lsid].y1) };
      if (e[0] != lsid) {
        if (e[0] != 0) sl_push(&als, whole);
        continue;
      }
      double x0 = ls[lsid].x0, y0 = ls[lsid].y0, x1 = ls[lsid].x1, y1 = ls[lsid].y1;
      if (!clip_to_box(&x0, &y0, &x1, &y1, iw - e[1], ih - e[3], e[2], e[4])) continue;
      const seg2 cl = { cvec2(x0, y0), cvec2(x1, y1) };
      sl_push(&als, cl);
    }
    rect_t r;
    if (funnel(&als, iw, ih, tanAOV, 0, &r)) PUSH_RECT(r);
  }
  for (int gi = 0; gi < ngroups; gi++) { intlist *set = (intlist *)ArrayMap_get(groups, keys[gi]); free(set->v); free(set); }
  free(keys);
  ArrayMap_dispose(groups);

  /* pass 2: every polyline on its own, long segments only */
  for (int i = 1; i <= n; i++) {
    if (ls[i].polyid == 0 || ls[i].leftPtr > 0) continue;
    seglist als = { 0 };
    for (int j = i; j > 0 && j <= n; j = ls[j].rightPtr) {
      const vec2 e0 = cvec2(ls[j].x0, ls[j].y0), e1 = cvec2(ls[j].x1, ls[j].y1);
      if (distanceSqu2(e0, e1) > 32.0 * 32.0) { const seg2 s = { e0, e1 }; sl_push(&als, s); }
    }
    rect_t r;
    if (funnel(&als, iw, ih, tanAOV, 2, &r)) PUSH_RECT(r);
  }

  ret[0].nItems = nret;
  return ret;
}

/* probes computed on the host from full planes, exactly like rh:1066-1098 does it */
void *rd_postprocess_planes(const void *segs, const int32_t *boundary, const int32_t *table, int iw, int ih, double tanAOV) {
  const linesegment_t *ls = (const linesegment_t *)segs;
  const int n = ((const int *)segs)[0];
  const unsigned nentry = (unsigned)(iw * ih * 4 / 5);
  int *probes = (int *)calloc((size_t)(n + 1) * 15 * 6, sizeof(int));
  for (int i = 1; i <= n; i++) {
    if (ls[i].polyid == 0) continue;
    const double x0 = rint(ls[i].x0), y0 = rint(ls[i].y0), x1 = rint(ls[i].x1), y1 = rint(ls[i].y1);
    const vec2 d = normalize2(minus2(cvec2(x1, y1), cvec2(x0, y0)));
    const vec2 vd = cvec2(-d.a[1], d.a[0]);
    for (int j = 0; j < 3; j++)
      for (int dist = -2; dist <= 2; dist++) {
        const vec2 p = plus2(cvec2(x0, y0), dot2(minus2(cvec2(x1, y1), cvec2(x0, y0)), (j + 0.5) / 3));
        const vec2 c = plus2(p, dot2(vd, dist));
        const int x = (int)(c.a[0] + 0.5), y = (int)(c.a[1] + 0.5);
        int *pr = probes + (size_t)(i * 15 + j * 5 + dist + 2) * 6;
        if (x < 0 || x >= iw || y < 0 || y >= ih) continue;
        const int segid = boundary[x + y * iw];
        pr[0] = segid;
        if (segid > 0) {
          const unsigned slot = (((uint32_t)i * (uint32_t)segid) & 0x7fffffffu) % nentry;
          for (int q = 0; q < 5; q++) pr[1 + q] = table[(size_t)slot * 5 + q];
        }
      }
  }
  void *r = rd_post_run(segs, n + 1, probes, iw, ih, tanAOV);
  free(probes);
  return r;
}
