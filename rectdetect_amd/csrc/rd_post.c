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

/* ------------------------------------------------------------------ pose of a quadrilateral (behaviour of rh:427-656)
 *
 * Four image corners define four unit viewing rays r_i; wanted are depths t_i such that the points P_i = t_i r_i form a planar
 * rectangle.  The reference minimises a 12-term residual twice - once for each way of pairing the sides with the unit
 * length ("pairing" 0 and 1) - by a diagonally preconditioned non-linear conjugate-gradient descent with finite-difference
 * derivatives, and keeps the better one.  The two descents do not depend on each other, so they are carried through the
 * iteration side by side here (index m; structure of arrays): on the host that is a loop of two, in a device port it is the
 * lane index.  What is fixed by the reference is the arithmetic - every sum below is evaluated in the
 * reference's operand order, because the rect_t doubles must match bit for bit - not the decomposition, which is ours.
 */
#define PAIRINGS 2
#define FD_STEP (1e-6)        /* finite-difference step (rh:430) */
#define CG_STEPS 12           /* rh:611,618 */
#define WALK_STEPS 10         /* Newton steps per line search (rh:611,618) */
#define CG_RESTART 10         /* steepest-descent restart period (rh:575) */

typedef struct { double r[4][3]; } rays4;

enum { S01, S12, S23, S03, S02, S13, NSIDES };                       /* squared distances between P_a and P_b */
static const int side_a[NSIDES] = { 0, 1, 2, 0, 0, 1 }, side_b[NSIDES] = { 1, 2, 3, 3, 2, 3 };
/* per pairing: the two sides that must have length 1, the side that scales the relative terms, and the corner that
 * takes part first in the two parallelogram defects (the other one is its opposite, 2 - first) */
static const int unit_p[PAIRINGS] = { S03, S23 }, unit_q[PAIRINGS] = { S12, S01 }, scale_s[PAIRINGS] = { S01, S12 }, first_c[PAIRINGS] = { 2, 0 };

static inline double sum_sq3(const double w[3]) { return w[0] * w[0] + w[1] * w[1] + w[2] * w[2]; }
static inline double inner3(const double u[3], const double w[3]) { return u[0] * w[0] + u[1] * w[1] + u[2] * w[2]; }
static inline void normal_of(const double P[4][3], int apex, int i, int j, double n[3]) {      /* (P_i - P_apex) x (P_j - P_apex) */
  double u[3], w[3];
  for (int c = 0; c < 3; c++) { u[c] = P[i][c] - P[apex][c]; w[c] = P[j][c] - P[apex][c]; }
  n[0] = u[1] * w[2] - u[2] * w[1]; n[1] = u[2] * w[0] - u[0] * w[2]; n[2] = u[0] * w[1] - u[1] * w[0];
}

/* the residual of pairing m at depths t (rh:442-477: two unit sides, two parallelogram defects, four right angles by
 * Pythagoras, two planarity terms; the relative terms are divided by the squared length of the free side) */
static double rect_defect(const rays4 *R, int m, const double t[4]) {
  double P[4][3], L[NSIDES];
  for (int i = 0; i < 4; i++) for (int c = 0; c < 3; c++) P[i][c] = R->r[i][c] * t[i];
  for (int k = 0; k < NSIDES; k++) {
    double d[3];
    for (int c = 0; c < 3; c++) d[c] = P[side_a[k]][c] - P[side_b[k]][c];
    L[k] = sum_sq3(d);
  }
  const int f = first_c[m], o = 2 - f;
  const double rel = 1.0 / L[scale_s[m]];
  double e[3], acc = 0;
  acc += sq(L[unit_p[m]] - 1);
  acc += sq(L[unit_q[m]] - 1);
  for (int c = 0; c < 3; c++) e[c] = (P[f][c] - P[1][c]) + (P[o][c] - P[3][c]);
  acc += sum_sq3(e);
  for (int c = 0; c < 3; c++) e[c] = (P[1][c] - P[o][c]) + (P[3][c] - P[f][c]);
  acc += rel * sum_sq3(e);
  acc += sq(L[S01] + L[S12] - L[S02]);
  acc += sq(L[S03] + L[S23] - L[S02]);
  acc += sq(L[S01] + L[S03] - L[S13]);
  acc += sq(L[S12] + L[S23] - L[S13]);
  double n[3];
  normal_of(P, 0, 1, 3, n);
  acc += rel * sq(inner3(n, P[2]) - inner3(n, P[0])) / inner3(n, n);
  normal_of(P, 1, 0, 2, n);
  acc += rel * sq(inner3(n, P[3]) - inner3(n, P[1])) / inner3(n, n);
  return acc;
}

/* descent state of the two pairings, side by side */
typedef struct {
  double t[PAIRINGS][4];        /* current depths */
  double res[PAIRINGS][4];      /* negative gradient */
  double pre[PAIRINGS][4];      /* preconditioned residual */
  double dir[PAIRINGS][4];      /* search direction */
  double rp[PAIRINGS];          /* res . pre of the current point */
  int since_restart[PAIRINGS];
} descent2;

static inline double inner4(const double u[4], const double w[4]) { return u[0] * w[0] + u[1] * w[1] + u[2] * w[2] + u[3] * w[3]; }

/* central differences along the four axes (rh:492-512) -> res = -gradient; pre = res / curvature per axis when every
 * curvature is positive, else res (rh:538-555) */
static void probe_axes(const rays4 *R, descent2 *D) {
  for (int m = 0; m < PAIRINGS; m++) {
    const double f0 = rect_defect(R, m, D->t[m]);
    double curv[4];
    int convex = 1;
    for (int i = 0; i < 4; i++) {
      double lo[4], hi[4];
      for (int j = 0; j < 4; j++) { const double h = j == i ? FD_STEP : 0; lo[j] = D->t[m][j] - h; hi[j] = D->t[m][j] + h; }
      const double fl = rect_defect(R, m, lo), fh = rect_defect(R, m, hi);
      D->res[m][i] = (fh - fl) / (2 * FD_STEP) * -1;
      curv[i] = (fl - 2 * f0 + fh) / (FD_STEP * FD_STEP);
      if (curv[i] <= 0) convex = 0;
    }
    for (int i = 0; i < 4; i++) {
      if (convex) { D->pre[m][i] = 1.0 / curv[i]; D->pre[m][i] *= D->res[m][i]; }
      else D->pre[m][i] = D->res[m][i];
    }
  }
}

/* WALK_STEPS damped Newton steps along dir (rh:514-536): slope and curvature from a three-point stencil, the step is halved
 * whenever it does not lower the residual, a pairing stops once its step falls below 1e-10 */
static void walk_along(const rays4 *R, descent2 *D) {
  double u[PAIRINGS][4], damp[PAIRINGS];
  int done[PAIRINGS];
  for (int m = 0; m < PAIRINGS; m++) {
    const double k = 1.0 / (sqrt(D->dir[m][0] * D->dir[m][0] + D->dir[m][1] * D->dir[m][1] + D->dir[m][2] * D->dir[m][2] + D->dir[m][3] * D->dir[m][3]) + 1e-20);
    for (int i = 0; i < 4; i++) u[m][i] = D->dir[m][i] * k;
    damp[m] = 1.0; done[m] = 0;
  }
  for (int step = 0; step < WALK_STEPS; step++)
    for (int m = 0; m < PAIRINGS; m++) {
      if (done[m]) continue;
      double fwd[4], bwd[4], cand[4];
      for (int i = 0; i < 4; i++) { fwd[i] = D->t[m][i] + u[m][i] * FD_STEP; bwd[i] = D->t[m][i] + u[m][i] * -FD_STEP; }
      const double f0 = rect_defect(R, m, D->t[m]), ff = rect_defect(R, m, fwd), fb = rect_defect(R, m, bwd);
      const double slope = (ff - fb) * (1.0 / (2 * FD_STEP));
      double curv = (ff + fb - 2 * f0) * (1.0 / (FD_STEP * FD_STEP));
      if (curv * curv < 1e-10) curv = 1;
      const double len = fabs(slope / curv);
      if (len < 1e-10) { done[m] = 1; continue; }
      for (int i = 0; i < 4; i++) cand[i] = D->t[m][i] + u[m][i] * (len * damp[m]);
      if (f0 < rect_defect(R, m, cand)) { damp[m] *= 0.5; continue; }
      for (int i = 0; i < 4; i++) D->t[m][i] = cand[i];
    }
}

/* rh:557-588: Polak-Ribiere conjugate gradients on the preconditioned residual, restarted every CG_RESTART steps and whenever
 * beta is not positive.  On return D->t holds the depths of both pairings. */
static void descend(const rays4 *R, descent2 *D) {
  probe_axes(R, D);
  for (int m = 0; m < PAIRINGS; m++) {
    memcpy(D->dir[m], D->pre[m], sizeof(D->dir[m]));
    D->rp[m] = inner4(D->res[m], D->dir[m]);
    D->since_restart[m] = 0;
  }
  for (int it = 0; it < CG_STEPS; it++) {
    walk_along(R, D);
    double old_pre[PAIRINGS][4];
    memcpy(old_pre, D->pre, sizeof(old_pre));
    probe_axes(R, D);
    for (int m = 0; m < PAIRINGS; m++) {
      const double before = D->rp[m];
      const double cross = inner4(D->res[m], old_pre[m]);
      D->rp[m] = inner4(D->res[m], D->pre[m]);
      const double beta = (D->rp[m] - cross) / before;
      if (D->since_restart[m] == CG_RESTART || beta <= 0 || before == 0) {
        memcpy(D->dir[m], D->pre[m], sizeof(D->dir[m]));
        D->since_restart[m] = 0;
      } else
        for (int i = 0; i < 4; i++) D->dir[m][i] = D->pre[m][i] + D->dir[m][i] * beta;
      D->since_restart[m]++;
    }
  }
}

static inline double gap3(const double a[3], const double b[3]) {
  const double d[3] = { a[0] - b[0], a[1] - b[1], a[2] - b[2] };
  return sqrt(sum_sq3(d));
}

/* rh:590-634: sides = the four sides in angular order (side i starts at corner i), centre = their length-weighted centroid.
 * The corner order of the result starts at the side whose outward normal points most upwards in the image. */
static rect_t estimate_pose(const seg2 *sides, vec2 centre, int iw, int ih, double tanAOV) {
  int first = 0;
  double lowest = 1e+100;
  for (int i = 0; i < 4; i++) {
    const vec2 along = normalize2(minus2(sides[i].e1, sides[i].e0));
    vec2 out = cvec2(-along.a[1], along.a[0]);
    if (vdot2(minus2(sides[i].e0, centre), out) < 0) out = dot2(out, -1);
    if (out.a[1] < lowest) { lowest = out.a[1]; first = i; }
  }
  rays4 R;
  const double focal = iw / 2 / tanAOV;                        /* (integer half width, like the reference) */
  for (int i = 0; i < 4; i++) {
    const vec2 c = sides[(i + first) & 3].e0;
    const vec3 v = normalize3(cvec3(c.a[0] - (iw / 2), -(c.a[1] - ih / 2), focal));
    for (int k = 0; k < 3; k++) R.r[i][k] = v.a[k];
  }
  /* starting depths: both ends of a unit side at the depth where that side subtends unit length (rh:607-617) */
  const double a01 = 1.0 / gap3(R.r[0], R.r[1]), a23 = 1.0 / gap3(R.r[2], R.r[3]);
  const double a12 = 1.0 / gap3(R.r[1], R.r[2]), a03 = 1.0 / gap3(R.r[0], R.r[3]);
  descent2 D;
  memset(&D, 0, sizeof(D));
  D.t[1][0] = a01; D.t[1][1] = a01; D.t[1][2] = a23; D.t[1][3] = a23;
  D.t[0][0] = a03; D.t[0][1] = a12; D.t[0][2] = a12; D.t[0][3] = a03;
  descend(&R, &D);
  const double f1 = rect_defect(&R, 1, D.t[1]), f0 = rect_defect(&R, 0, D.t[0]);
  const int best = f1 < f0 ? 1 : 0;

  rect_t ret;
  memset(&ret, 0, sizeof(ret));
  ret.value = best ? f1 : f0;
  const double flip = D.t[best][0] < 0 ? -1 : 1;               /* a mirrored solution behind the camera is turned round */
  for (int i = 0; i < 4; i++) {
    const double depth = flip < 0 ? D.t[best][i] * -1 : D.t[best][i];
    for (int k = 0; k < 3; k++) ret.c3[i].a[k] = R.r[i][k] * depth;
    ret.c2[i] = sides[(i + first) & 3].e0;
  }
  return ret;
}

/* rh:636-656: small residual, in front of the camera, aspect ratio within 1:12, and no corner much closer to a side's
 * segment than the farthest one is (ratio of squared distances <= 100) */
static int looks_like_a_screen(const rect_t *r) {
  if (r->value > 0.05) return 0;
  for (int i = 0; i < 4; i++) if (r->c3[i].a[2] < 0) return 0;
  const double aspect = distance3(r->c3[0], r->c3[1]) / distance3(r->c3[1], r->c3[2]);
  if (aspect < 1.0 / 12 || 12 < aspect) return 0;
  double widest = 0, narrowest = 1e+100;
  for (int i = 0; i < 4; i++) {
    double reach[2];
    for (int k = 0; k < 2; k++) {
      const vec2 far = r->c2[(i + 2 + k) % 4];
      reach[k] = distanceSqu2(far, closest_on_seg(r->c2[i], r->c2[(i + 1) % 4], far));
    }
    const double m = fmax(reach[0], reach[1]);
    widest = fmax(widest, m);
    narrowest = fmin(narrowest, m);
  }
  return widest / narrowest > 100 ? 0 : 1;
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
