/*
 * rectdetect-mi355x: line segments -> rectangles, the arithmetic shared by the host post-process (rd_post.c, C) and the device
 * post-process (rd_k_post.hip, one wave per candidate): candidate funnel and pose estimation.
 *
 * Behaviour restated from the reference's executeCPUTask helpers (oclrect.c:385-1045, "rh"); every sum is evaluated in the
 * reference's operand order, in double precision, without FMA contraction (see rd_device.h), because the rect_t doubles are
 * compared bit for bit with the reference's.  What is ours is the shape: no allocation anywhere (fixed-capacity work space handed
 * in by the caller, overflow reported), the hull without recursion, the two pose descents side by side (index m: a loop on the
 * host, a lane on the device).  Compiles as C and as HIP C++.
 */
#ifndef RD_POST_CORE_H
#define RD_POST_CORE_H
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define RD_HD __host__ __device__ static inline
#else
#define RD_HD static inline
#endif

typedef struct { double x, y; } rdp_p2;
typedef struct { rdp_p2 e0, e1; } rdp_seg;
typedef struct { double c2[4][2]; double c3[4][3]; double value; uint32_t status, pad; } rdp_rect;      /* = rect_t (oclrect.h), 176 bytes */

/* one pending call of the hull construction (see rdp_hull_side) */
typedef struct { int s0, sn, l0, ln, mark, stage; rdp_p2 left, right, pf; } rdp_hull_frame;
#define RDP_HULL_DEPTH 48

/* work space of one candidate: cap segments in `als` and `out`, 2*cap points, hull points, 16*cap pool ints and the hull's call stack */
typedef struct {
  rdp_seg *als, *out;
  rdp_p2 *pts, *hull;
  int *pool;
  rdp_hull_frame *stack;
  int cap, overflow;
} rdp_work;
#define RDP_WORK_BYTES(cap) ((size_t)(cap) * (2 * sizeof(rdp_seg) + 4 * sizeof(rdp_p2) + 16 * sizeof(int)) + RDP_HULL_DEPTH * sizeof(rdp_hull_frame))
RD_HD void rdp_work_place(rdp_work *w, void *mem, int cap) {
  char *p = (char *)mem;
  w->stack = (rdp_hull_frame *)p; p += sizeof(rdp_hull_frame) * RDP_HULL_DEPTH;
  w->als = (rdp_seg *)p; p += sizeof(rdp_seg) * (size_t)cap;
  w->out = (rdp_seg *)p; p += sizeof(rdp_seg) * (size_t)cap;
  w->pts = (rdp_p2 *)p; p += sizeof(rdp_p2) * (size_t)cap * 2;
  w->hull = (rdp_p2 *)p; p += sizeof(rdp_p2) * (size_t)cap * 2;
  w->pool = (int *)p;
  w->cap = cap; w->overflow = 0;
}

/* ------------------------------------------------------------------ 2-D geometry (rh:389-425) */
RD_HD double rdp_sq(double v) { return v * v; }
RD_HD rdp_p2 rdp_pt(double x, double y) { rdp_p2 p; p.x = x; p.y = y; return p; }
RD_HD double rdp_d2(rdp_p2 p, rdp_p2 q) { const double dx = p.x - q.x, dy = p.y - q.y; return dx * dx + dy * dy; }
RD_HD double rdp_dot(rdp_p2 p, rdp_p2 q) { return p.x * q.x + p.y * q.y; }
RD_HD rdp_p2 rdp_sub(rdp_p2 p, rdp_p2 q) { return rdp_pt(p.x - q.x, p.y - q.y); }
RD_HD rdp_p2 rdp_unit(rdp_p2 p) { const double k = 1.0 / (sqrt(p.x * p.x + p.y * p.y) + 1e-20); return rdp_pt(p.x * k, p.y * k); }
RD_HD float rdp_sqlen(const rdp_seg *s) { return (float)rdp_d2(s->e0, s->e1); }      /* rh:390: narrowed to float */

/* Probe k (0..14) of a segment (rh:1066-1098): point k / 5 of three along the segment between its end points rounded to integers, moved
 * k % 5 - 2 pixels along the segment's normal; the pixel it falls on goes to (*sx, *sy).  Returns 0 when that lies outside the frame.
 * Shared by the sampling kernel (rd_k_rect.hip: k_sample_segments) and rd_postprocess_planes. */
RD_HD int rdp_probe_pixel(float fx0, float fy0, float fx1, float fy1, int k, int iw, int ih, int *sx, int *sy) {
  const rdp_p2 a = rdp_pt(rint((double)fx0), rint((double)fy0)), b = rdp_pt(rint((double)fx1), rint((double)fy1));
  const rdp_p2 e = rdp_sub(b, a), u = rdp_unit(e);
  const double f = (k / 5 + 0.5) / 3;
  const int off = k % 5 - 2;
  const double cx = (a.x + e.x * f) + -u.y * off, cy = (a.y + e.y * f) + u.x * off;
  *sx = (int)(cx + 0.5); *sy = (int)(cy + 0.5);
  return !(*sx < 0 || *sx >= iw || *sy < 0 || *sy >= ih);
}

/* foot of the perpendicular from p on the LINE through v, w (rh:400-406) */
RD_HD rdp_p2 rdp_foot(rdp_p2 v, rdp_p2 w, rdp_p2 p) {
  const double l2 = rdp_d2(v, w);
  if (l2 == 0.0) return v;
  const double t = ((p.x - v.x) * (w.x - v.x) + (p.y - v.y) * (w.y - v.y)) / l2;
  return rdp_pt(v.x + t * (w.x - v.x), v.y + t * (w.y - v.y));
}
/* closest point of the SEGMENT v-w to p (rh:408-416) */
RD_HD rdp_p2 rdp_closest(rdp_p2 v, rdp_p2 w, rdp_p2 p) {
  const double l2 = rdp_d2(v, w);
  if (l2 == 0.0) return v;
  const double t = ((p.x - v.x) * (w.x - v.x) + (p.y - v.y) * (w.y - v.y)) / l2;
  if (t < 0) return v;
  if (t > 1.0) return w;
  return rdp_pt(v.x + t * (w.x - v.x), v.y + t * (w.y - v.y));
}
/* intersection of the lines through u and v; returns 0 when (nearly) parallel (rh:418-425) */
RD_HD int rdp_cross_lines(const rdp_seg *u, const rdp_seg *v, rdp_p2 *at) {
  const double d = (v->e1.x - v->e0.x) * (u->e1.y - u->e0.y) - (v->e1.y - v->e0.y) * (u->e1.x - u->e0.x);
  if (fabs(d) < 1e-4) return 0;
  const double n = (v->e0.y - u->e0.y) * (u->e1.x - u->e0.x) - (v->e0.x - u->e0.x) * (u->e1.y - u->e0.y);
  const double q = n / d;
  *at = rdp_pt(v->e0.x + q * (v->e1.x - v->e0.x), v->e0.y + q * (v->e1.y - v->e0.y));
  return 1;
}

/* ------------------------------------------------------------------ pose of a quadrilateral (behaviour of rh:427-656)
 *
 * Four image corners define four unit viewing rays r_i; wanted are depths t_i such that the points P_i = t_i r_i form a planar
 * rectangle.  The reference minimises a 12-term residual twice - once for each way of pairing the sides with the unit length
 * ("pairing" 0 and 1) - by a diagonally preconditioned non-linear conjugate-gradient descent with finite-difference derivatives,
 * and keeps the better one.  rdp_defect is that residual; the descent lives with its caller (serial on the host, rd_post.c; the
 * evaluations of a stencil spread over lanes on the device, rd_k_post.hip).
 */
#define RDP_PAIRINGS 2
#define RDP_FD_STEP (1e-6)        /* finite-difference step (rh:430) */
#define RDP_CG_STEPS 12           /* rh:611,618 */
#define RDP_WALK_STEPS 10         /* Newton steps per line search (rh:611,618) */
#define RDP_CG_RESTART 10         /* steepest-descent restart period (rh:575) */

typedef struct { double r[4][3]; } rdp_rays;

RD_HD double rdp_sum_sq3(const double *w) { return w[0] * w[0] + w[1] * w[1] + w[2] * w[2]; }
RD_HD double rdp_inner3(const double *u, const double *w) { return u[0] * w[0] + u[1] * w[1] + u[2] * w[2]; }
RD_HD double rdp_inner4(const double *u, const double *w) { return u[0] * w[0] + u[1] * w[1] + u[2] * w[2] + u[3] * w[3]; }

/* the residual of pairing m at depths t (rh:442-477: two unit sides, two parallelogram defects, four right angles by Pythagoras,
 * two planarity terms; the relative terms are divided by the squared length of the free side).
 * Squared distances: 0 = |P0 P1|, 1 = |P1 P2|, 2 = |P2 P3|, 3 = |P0 P3|, 4 = |P0 P2|, 5 = |P1 P3|. */
RD_HD double rdp_defect(const rdp_rays *R, int m, const double *t) {
  double P[4][3], L[6];
  for (int i = 0; i < 4; i++) for (int c = 0; c < 3; c++) P[i][c] = R->r[i][c] * t[i];
  for (int k = 0; k < 6; k++) {
    const int a = k == 1 ? 1 : (k == 2 ? 2 : (k == 5 ? 1 : 0)), b = k == 0 ? 1 : (k == 1 || k == 4 ? 2 : 3);
    double d[3];
    for (int c = 0; c < 3; c++) d[c] = P[a][c] - P[b][c];
    L[k] = rdp_sum_sq3(d);
  }
  /* per pairing: the two sides that must have length 1, the side that scales the relative terms, the corner that takes part first
   * in the two parallelogram defects (the other one is its opposite) */
  const int f = m ? 0 : 2, o = 2 - f;
  const double rel = 1.0 / L[m ? 1 : 0];
  double e[3], n[3], u[3], w[3], acc = 0;
  acc += rdp_sq(L[m ? 2 : 3] - 1);
  acc += rdp_sq(L[m ? 0 : 1] - 1);
  for (int c = 0; c < 3; c++) e[c] = (P[f][c] - P[1][c]) + (P[o][c] - P[3][c]);
  acc += rdp_sum_sq3(e);
  for (int c = 0; c < 3; c++) e[c] = (P[1][c] - P[o][c]) + (P[3][c] - P[f][c]);
  acc += rel * rdp_sum_sq3(e);
  acc += rdp_sq(L[0] + L[1] - L[4]);
  acc += rdp_sq(L[3] + L[2] - L[4]);
  acc += rdp_sq(L[0] + L[3] - L[5]);
  acc += rdp_sq(L[1] + L[2] - L[5]);
  for (int c = 0; c < 3; c++) { u[c] = P[1][c] - P[0][c]; w[c] = P[3][c] - P[0][c]; }
  n[0] = u[1] * w[2] - u[2] * w[1]; n[1] = u[2] * w[0] - u[0] * w[2]; n[2] = u[0] * w[1] - u[1] * w[0];
  acc += rel * rdp_sq(rdp_inner3(n, P[2]) - rdp_inner3(n, P[0])) / rdp_inner3(n, n);
  for (int c = 0; c < 3; c++) { u[c] = P[0][c] - P[1][c]; w[c] = P[2][c] - P[1][c]; }
  n[0] = u[1] * w[2] - u[2] * w[1]; n[1] = u[2] * w[0] - u[0] * w[2]; n[2] = u[0] * w[1] - u[1] * w[0];
  acc += rel * rdp_sq(rdp_inner3(n, P[3]) - rdp_inner3(n, P[1])) / rdp_inner3(n, n);
  return acc;
}

/* rh:590-617: sides = the four sides in angular order (side i starts at corner i), centre = their length-weighted centroid.  The
 * corner order of the result starts at the side whose outward normal points most upwards in the image.  Fills the rays, the index of
 * the first side and the starting depths of both pairings (both ends of a unit side at the depth where that side subtends unit length). */
RD_HD void rdp_pose_setup(const rdp_seg *sides, rdp_p2 centre, int iw, int ih, double tanAOV, rdp_rays *R, int *first_out, double (*t0)[4]) {
  int first = 0;
  double lowest = 1e+100;
  for (int i = 0; i < 4; i++) {
    const rdp_p2 along = rdp_unit(rdp_sub(sides[i].e1, sides[i].e0));
    rdp_p2 outw = rdp_pt(-along.y, along.x);
    if (rdp_dot(rdp_sub(sides[i].e0, centre), outw) < 0) outw = rdp_pt(outw.x * -1, outw.y * -1);
    if (outw.y < lowest) { lowest = outw.y; first = i; }
  }
  const double focal = iw / 2 / tanAOV;                        /* (integer half width, like the reference) */
  for (int i = 0; i < 4; i++) {
    const rdp_p2 c = sides[(i + first) & 3].e0;
    const double v[3] = { c.x - (iw / 2), -(c.y - ih / 2), focal };
    const double k = 1.0 / (sqrt(rdp_sum_sq3(v)) + 1e-20);
    for (int q = 0; q < 3; q++) R->r[i][q] = v[q] * k;
  }
  double g[4];          /* 1 / |r_a - r_b| for the sides 01, 23, 12, 03 */
  for (int k = 0; k < 4; k++) {
    const int a = k == 1 ? 2 : (k == 2 ? 1 : 0), b = k == 0 ? 1 : (k == 2 ? 2 : 3);
    const double d[3] = { R->r[a][0] - R->r[b][0], R->r[a][1] - R->r[b][1], R->r[a][2] - R->r[b][2] };
    g[k] = 1.0 / sqrt(rdp_sum_sq3(d));
  }
  t0[1][0] = g[0]; t0[1][1] = g[0]; t0[1][2] = g[1]; t0[1][3] = g[1];
  t0[0][0] = g[3]; t0[0][1] = g[2]; t0[0][2] = g[2]; t0[0][3] = g[3];
  *first_out = first;
}

/* rh:619-656: the better pairing's depths -> rect_t, and the "looks like a screen" bit: small residual, in front of the camera, aspect
 * ratio within 1:12, no corner much closer to a side's segment than the farthest one is (ratio of squared distances <= 100) */
RD_HD void rdp_pose_finish(const rdp_seg *sides, int first, const rdp_rays *R, const double (*t)[4], const double *f, uint32_t status, rdp_rect *ret) {
  const int best = f[1] < f[0] ? 1 : 0;
  ret->value = f[best];
  const int flip = t[best][0] < 0;                             /* a mirrored solution behind the camera is turned round */
  for (int i = 0; i < 4; i++) {
    const double depth = flip ? t[best][i] * -1 : t[best][i];
    for (int k = 0; k < 3; k++) ret->c3[i][k] = R->r[i][k] * depth;
    ret->c2[i][0] = sides[(i + first) & 3].e0.x; ret->c2[i][1] = sides[(i + first) & 3].e0.y;
  }
  ret->status = status; ret->pad = 0;
  if (ret->value > 0.05) return;
  for (int i = 0; i < 4; i++) if (ret->c3[i][2] < 0) return;
  double d01[3], d12[3];
  for (int k = 0; k < 3; k++) { d01[k] = ret->c3[0][k] - ret->c3[1][k]; d12[k] = ret->c3[1][k] - ret->c3[2][k]; }
  const double aspect = sqrt(rdp_sum_sq3(d01)) / sqrt(rdp_sum_sq3(d12));
  if (aspect < 1.0 / 12 || 12 < aspect) return;
  double widest = 0, narrowest = 1e+100;
  for (int i = 0; i < 4; i++) {
    double reach[2];
    const rdp_p2 a = rdp_pt(ret->c2[i][0], ret->c2[i][1]), b = rdp_pt(ret->c2[(i + 1) % 4][0], ret->c2[(i + 1) % 4][1]);
    for (int k = 0; k < 2; k++) {
      const rdp_p2 far_c = rdp_pt(ret->c2[(i + 2 + k) % 4][0], ret->c2[(i + 2 + k) % 4][1]);
      reach[k] = rdp_d2(far_c, rdp_closest(a, b, far_c));
    }
    const double mx = fmax(reach[0], reach[1]);
    widest = fmax(widest, mx);
    narrowest = fmin(narrowest, mx);
  }
  if (!(widest / narrowest > 100)) ret->status |= 1;
}

/* ------------------------------------------------------------------ convex hull (behaviour of rh:658-734, quick hull)
 * Same vertex order as the reference's recursion - right-most point, the points above the line left-right from right to left, left-most
 * point, the points below - with an explicit stack; subsets are index lists in w->pool, released in LIFO order. */
RD_HD int rdp_hull_side(rdp_work *w, int npts, int nh, int s0, int sn, rdp_p2 left, rdp_p2 right, int *pool_top) {
  rdp_hull_frame *st = w->stack;
  int sp = 0;
  st[0].s0 = s0; st[0].sn = sn; st[0].left = left; st[0].right = right; st[0].stage = 0; st[0].mark = *pool_top; st[0].l0 = 0; st[0].ln = 0; st[0].pf = left;
  while (sp >= 0) {
    rdp_hull_frame *f = &st[sp];
    if (f->stage == 0) {
      int far_i = -1;
      double d = 0;
      for (int i = 0; i < f->sn; i++) {
        const rdp_p2 p = w->pts[w->pool[f->s0 + i]];
        const double e = rdp_d2(rdp_foot(f->left, f->right, p), p);
        if (far_i < 0 || e > d) { far_i = i; d = e; }
      }
      if (d < 0.01 || far_i < 0) { *pool_top = f->mark; sp--; continue; }
      const rdp_p2 pf = w->pts[w->pool[f->s0 + far_i]];
      const rdp_p2 nr = rdp_pt(pf.y - f->right.y, f->right.x - pf.x);
      const rdp_p2 nl = rdp_pt(f->left.y - pf.y, pf.x - f->left.x);
      if (*pool_top + 2 * f->sn > 16 * w->cap || sp + 1 >= RDP_HULL_DEPTH) { w->overflow = 1; return nh; }
      const int r0 = *pool_top;
      int rn = 0;
      for (int i = 0; i < f->sn; i++) { if (i == far_i) continue; const int q = w->pool[f->s0 + i]; if (rdp_dot(rdp_sub(w->pts[q], pf), nr) > 0) w->pool[r0 + rn++] = q; }
      const int l0 = r0 + rn;
      int ln = 0;
      for (int i = 0; i < f->sn; i++) { if (i == far_i) continue; const int q = w->pool[f->s0 + i]; if (rdp_dot(rdp_sub(w->pts[q], pf), nl) > 0) w->pool[l0 + ln++] = q; }
      *pool_top = l0 + ln;
      f->pf = pf; f->l0 = l0; f->ln = ln; f->stage = 1;
      sp++;
      st[sp].s0 = r0; st[sp].sn = rn; st[sp].left = pf; st[sp].right = f->right; st[sp].stage = 0; st[sp].mark = *pool_top; st[sp].l0 = 0; st[sp].ln = 0; st[sp].pf = pf;
    } else if (f->stage == 1) {
      if (nh >= 2 * w->cap) { w->overflow = 1; return nh; }
      w->hull[nh++] = f->pf;
      f->stage = 2;
      sp++;
      st[sp].s0 = f->l0; st[sp].sn = f->ln; st[sp].left = f->left; st[sp].right = f->pf; st[sp].stage = 0; st[sp].mark = *pool_top; st[sp].l0 = 0; st[sp].ln = 0; st[sp].pf = f->pf;
    } else { *pool_top = f->mark; sp--; }
  }
  (void)npts;
  return nh;
}

/* hull of w->pts[0..npts) into w->hull; returns the number of hull points */
RD_HD int rdp_hull(rdp_work *w, int npts) {
  if (npts == 0) return 0;
  rdp_p2 right = w->pts[0], left = w->pts[0];
  for (int i = 0; i < npts; i++) {
    if (w->pts[i].x > right.x) right = w->pts[i];
    if (w->pts[i].x < left.x) left = w->pts[i];
  }
  const rdp_p2 up = rdp_pt(left.y - right.y, right.x - left.x);
  int nt = 0, nb = 0;          /* top subset at pool[0..nt), bottom subset at pool[2*cap .. ) */
  const int b0 = 2 * w->cap;
  for (int i = 0; i < npts; i++) {
    const rdp_p2 p = w->pts[i];
    if (p.x == left.x && p.y == left.y) continue;
    if (p.x == right.x && p.y == right.y) continue;
    if (rdp_dot(rdp_sub(p, left), up) > 0) w->pool[nt++] = i; else w->pool[b0 + nb++] = i;
  }
  int top = 4 * w->cap, nh = 0;
  w->hull[nh++] = right;
  nh = rdp_hull_side(w, npts, nh, 0, nt, left, right, &top);
  if (w->overflow) return nh;
  w->hull[nh++] = left;
  nh = rdp_hull_side(w, npts, nh, b0, nb, right, left, &top);
  return nh;
}

/* ------------------------------------------------------------------ Cohen-Sutherland clip (rh:744-802) */
RD_HD int rdp_outcode(double x, double y, double xmin, double ymin, double xmax, double ymax) {
  int c = 0;
  if (x < xmin) c |= 1;
  if (x > xmax) c |= 2;
  if (y < ymin) c |= 4;
  if (y > ymax) c |= 8;
  return c;
}
RD_HD int rdp_clip(double *x0, double *y0, double *x1, double *y1, double xmin, double ymin, double xmax, double ymax) {
  int c0 = rdp_outcode(*x0, *y0, xmin, ymin, xmax, ymax), c1 = rdp_outcode(*x1, *y1, xmin, ymin, xmax, ymax);
  for (;;) {
    if ((c0 | c1) == 0) return 1;
    if ((c0 & c1) != 0) return 0;
    double x = 0, y = 0;
    const int co = c0 != 0 ? c0 : c1;
    if (co & 8) { x = *x0 + (*x1 - *x0) * (ymax - *y0) / (*y1 - *y0); y = ymax; }
    else if (co & 4) { x = *x0 + (*x1 - *x0) * (ymin - *y0) / (*y1 - *y0); y = ymin; }
    else if (co & 2) { y = *y0 + (*y1 - *y0) * (xmax - *x0) / (*x1 - *x0); x = xmax; }
    else if (co & 1) { y = *y0 + (*y1 - *y0) * (xmin - *x0) / (*x1 - *x0); x = xmin; }
    if (co == c0) { *x0 = x; *y0 = y; c0 = rdp_outcode(*x0, *y0, xmin, ymin, xmax, ymax); }
    else { *x1 = x; *y1 = y; c1 = rdp_outcode(*x1, *y1, xmin, ymin, xmax, ymax); }
  }
}

/* ------------------------------------------------------------------ the candidate funnel (rh:806-1045) */
/* stable ascending sort by the float squared length (the reference uses glibc qsort, a stable merge sort) */
RD_HD void rdp_sort_by_length(rdp_seg *v, int n) {
  for (int i = 1; i < n; i++) {
    const rdp_seg k = v[i];
    const float kl = rdp_sqlen(&k);
    int j = i - 1;
    while (j >= 0 && rdp_sqlen(&v[j]) > kl) { v[j + 1] = v[j]; j--; }
    v[j + 1] = k;
  }
}
RD_HD double rdp_outward_angle(const rdp_seg *s, rdp_p2 c) {
  rdp_p2 v = rdp_sub(s->e0, s->e1);
  v = rdp_pt(v.y, -v.x);
  if (rdp_dot(v, rdp_sub(s->e0, c)) < 0) v = rdp_pt(v.x * -1, v.y * -1);
  return atan2(v.x, v.y);
}
/* rh:821-852: stable ascending sort by the direction of the outward normal */
RD_HD void rdp_sort_by_angle(rdp_seg *v, int n, rdp_p2 c) {
  for (int i = 1; i < n; i++) {
    const rdp_seg k = v[i];
    const double ka = rdp_outward_angle(&k, c);
    int j = i - 1;
    while (j >= 0 && rdp_outward_angle(&v[j], c) > ka) { v[j + 1] = v[j]; j--; }
    v[j + 1] = k;
  }
}
/* rh:864-877 */
RD_HD rdp_p2 rdp_weighted_centre(const rdp_seg *v, int n) {
  rdp_p2 g = rdp_pt(0, 0);
  double sum = 0;
  for (int i = 0; i < n; i++) {
    const double len = sqrt(rdp_d2(v[i].e0, v[i].e1));
    g = rdp_pt(g.x + (v[i].e0.x + v[i].e1.x) * len, g.y + (v[i].e0.y + v[i].e1.y) * len);
    sum += len;
  }
  const double k = 0.5 / sum;
  return rdp_pt(g.x * k, g.y * k);
}
/* rh:879-884 */
RD_HD double rdp_total_length(const rdp_seg *v, int n) {
  double r = 0;
  for (int i = 0; i < n; i++) r += sqrt((double)rdp_sqlen(&v[i]));      /* (the float squared length, widened: rh:882) */
  return r;
}

/* rh:1134-1160 / rh:1190-1216: the candidate's segments w->als[0..n) -> four sides in angular order (in w->out[0..4)) and their
 * centre; returns 1 when the candidate survives: short segments dropped (rh:926-943), only segments on the hull kept (rh:945-992),
 * the four longest (rh:994-1009) turned into corners (rh:1011-1045), not nearly a triangle (rh:886-895), convex (rh:897-922). */
RD_HD int rdp_funnel(rdp_work *w, int n, rdp_p2 *centre_out) {
  rdp_seg *in = w->als;
  /* drop_short: while more than four, the shortest goes if it is below 5 % of the longest (squared: 0.0025) */
  if (n > 4) {
    rdp_sort_by_length(in, n);
    const float longest = rdp_sqlen(&in[n - 1]);
    int first = 0;
    while (n - first > 4) {
      const float shortest = rdp_sqlen(&in[first]);
      if (shortest / longest > 0.05f * 0.05f) break;
      first++;
    }
    if (first > 0) { for (int i = first; i < n; i++) in[i - first] = in[i]; n -= first; }
  }
  /* keep_outer: for every hull edge the longest segment lying on it */
  if (2 * n > 2 * w->cap) { w->overflow = 1; return 0; }
  for (int i = 0; i < n; i++) { w->pts[2 * i] = in[i].e0; w->pts[2 * i + 1] = in[i].e1; }
  const int nh = rdp_hull(w, 2 * n);
  if (w->overflow) return 0;
  rdp_seg *out = w->out;
  int no = 0;
  for (int i = 0; i < nh; i++) {
    const rdp_p2 q0 = w->hull[i], q1 = w->hull[(i + 1) % nh];
    const rdp_p2 mid = rdp_pt((q0.x + q1.x) * 0.5, (q0.y + q1.y) * 0.5), nq = rdp_unit(rdp_sub(q0, q1));
    int added = -1;
    rdp_sort_by_length(in, n);
    for (int j = n - 1; j >= 0; j--) {
      const rdp_seg e = in[j];
      if (rdp_d2(mid, rdp_closest(e.e0, e.e1, mid)) < 1) { added = j; break; }
      if (fabs(rdp_dot(nq, rdp_unit(rdp_sub(e.e0, e.e1)))) > 0.95 && rdp_d2(mid, rdp_closest(e.e0, e.e1, mid)) / rdp_d2(q0, q1) < 0.01) { added = j; break; }
    }
    if (added != -1) {
      out[no++] = in[added];
      for (int k = added; k < n - 1; k++) in[k] = in[k + 1];
      n--;
    }
  }
  const double len0 = rdp_total_length(out, no);
  /* keep_longest(4): the four longest, longest first */
  if (no > 4) {
    rdp_sort_by_length(out, no);
    for (int k = 0; k < 4; k++) in[k] = out[no - 1 - k];
    for (int k = 0; k < 4; k++) out[k] = in[k];
    no = 4;
  }
  int ok = 1;
  if (no > 0) {
    rdp_sort_by_angle(out, no, rdp_weighted_centre(out, no));
    /* consecutive sides -> corners -> sides between corners; fails if two consecutive sides are parallel */
    rdp_p2 c[4];
    for (int i = 0; i < no && ok; i++) ok = rdp_cross_lines(&out[i], &out[(i + 1) % no], &c[i]);
    if (ok) for (int i = 0; i < no; i++) { out[i].e0 = c[i]; out[i].e1 = c[(i + 1) % no]; }
  }
  /* (with no segment left the reference still goes on: findCorners returns an empty list, the size test rejects it) */
  const double len1 = ok ? rdp_total_length(out, no) : 0;
  if (!ok) return 0;
  for (int i = 0; i < no; i++) {          /* nearly a triangle: a corner lies (almost) on the line through its neighbours */
    const rdp_seg a = out[i], b = out[(i + 1) % no];
    const double d0 = rdp_d2(a.e1, rdp_foot(a.e0, b.e1, a.e1)), d1 = rdp_d2(a.e0, b.e1);
    if (d0 / d1 < 0.001) return 0;
  }
  if (no < 4 || len1 / len0 > 2) return 0;
  int sign = 0;
  for (int i = 0; i < no; i++) {          /* convex: the turns all have one sign */
    const rdp_seg a = out[i], b = out[(i + 1) % no];
    const double ax = a.e1.x - a.e0.x, ay = a.e1.y - a.e0.y, bx = b.e1.x - b.e0.x, by = b.e1.y - b.e0.y;
    const int sg = ax * by - ay * bx > 0;
    if (i == 0) sign = sg; else if (sg != sign) return 0;
  }
  *centre_out = rdp_weighted_centre(out, no);
  return 1;
}

#endif
