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

#include "rd_post_core.h"

/* ------------------------------------------------------------------ the descent of both pairings, serially (rh:479-588)
 * (the device version, rd_k_post.hip, spreads the residual evaluations of a stencil over the lanes of a wave; the arithmetic of a
 *  step is the same) */
typedef struct {
  double t[RDP_PAIRINGS][4];        /* current depths */
  double res[RDP_PAIRINGS][4];      /* negative gradient */
  double pre[RDP_PAIRINGS][4];      /* preconditioned residual */
  double dir[RDP_PAIRINGS][4];      /* search direction */
  double rp[RDP_PAIRINGS];          /* res . pre of the current point */
  int since_restart[RDP_PAIRINGS];
} descent2;

/* central differences along the four axes (rh:492-512) -> res = -gradient; pre = res / curvature per axis when every
 * curvature is positive, else res (rh:538-555) */
static void probe_axes(const rdp_rays *R, descent2 *D) {
  for (int m = 0; m < RDP_PAIRINGS; m++) {
    const double f0 = rdp_defect(R, m, D->t[m]);
    double curv[4];
    int convex = 1;
    for (int i = 0; i < 4; i++) {
      double lo[4], hi[4];
      for (int j = 0; j < 4; j++) { const double h = j == i ? RDP_FD_STEP : 0; lo[j] = D->t[m][j] - h; hi[j] = D->t[m][j] + h; }
      const double fl = rdp_defect(R, m, lo), fh = rdp_defect(R, m, hi);
      D->res[m][i] = (fh - fl) / (2 * RDP_FD_STEP) * -1;
      curv[i] = (fl - 2 * f0 + fh) / (RDP_FD_STEP * RDP_FD_STEP);
      if (curv[i] <= 0) convex = 0;
    }
    for (int i = 0; i < 4; i++) {
      if (convex) { D->pre[m][i] = 1.0 / curv[i]; D->pre[m][i] *= D->res[m][i]; }
      else D->pre[m][i] = D->res[m][i];
    }
  }
}

/* RDP_WALK_STEPS damped Newton steps along dir (rh:514-536): slope and curvature from a three-point stencil, the step is halved
 * whenever it does not lower the residual, a pairing stops once its step falls below 1e-10 */
static void walk_along(const rdp_rays *R, descent2 *D) {
  double u[RDP_PAIRINGS][4], damp[RDP_PAIRINGS];
  int done[RDP_PAIRINGS];
  for (int m = 0; m < RDP_PAIRINGS; m++) {
    const double k = 1.0 / (sqrt(D->dir[m][0] * D->dir[m][0] + D->dir[m][1] * D->dir[m][1] + D->dir[m][2] * D->dir[m][2] + D->dir[m][3] * D->dir[m][3]) + 1e-20);
    for (int i = 0; i < 4; i++) u[m][i] = D->dir[m][i] * k;
    damp[m] = 1.0; done[m] = 0;
  }
  for (int step = 0; step < RDP_WALK_STEPS; step++)
    for (int m = 0; m < RDP_PAIRINGS; m++) {
      if (done[m]) continue;
      double fwd[4], bwd[4], cand[4];
      for (int i = 0; i < 4; i++) { fwd[i] = D->t[m][i] + u[m][i] * RDP_FD_STEP; bwd[i] = D->t[m][i] + u[m][i] * -RDP_FD_STEP; }
      const double f0 = rdp_defect(R, m, D->t[m]), ff = rdp_defect(R, m, fwd), fb = rdp_defect(R, m, bwd);
      const double slope = (ff - fb) * (1.0 / (2 * RDP_FD_STEP));
      double curv = (ff + fb - 2 * f0) * (1.0 / (RDP_FD_STEP * RDP_FD_STEP));
      if (curv * curv < 1e-10) curv = 1;
      const double len = fabs(slope / curv);
      if (len < 1e-10) { done[m] = 1; continue; }
      for (int i = 0; i < 4; i++) cand[i] = D->t[m][i] + u[m][i] * (len * damp[m]);
      if (f0 < rdp_defect(R, m, cand)) { damp[m] *= 0.5; continue; }
      for (int i = 0; i < 4; i++) D->t[m][i] = cand[i];
    }
}

/* rh:557-588: Polak-Ribiere conjugate gradients on the preconditioned residual, restarted every RDP_CG_RESTART steps and whenever
 * beta is not positive.  On return D->t holds the depths of both pairings. */
static void descend(const rdp_rays *R, descent2 *D) {
  probe_axes(R, D);
  for (int m = 0; m < RDP_PAIRINGS; m++) {
    memcpy(D->dir[m], D->pre[m], sizeof(D->dir[m]));
    D->rp[m] = rdp_inner4(D->res[m], D->dir[m]);
    D->since_restart[m] = 0;
  }
  for (int it = 0; it < RDP_CG_STEPS; it++) {
    walk_along(R, D);
    double old_pre[RDP_PAIRINGS][4];
    memcpy(old_pre, D->pre, sizeof(old_pre));
    probe_axes(R, D);
    for (int m = 0; m < RDP_PAIRINGS; m++) {
      const double before = D->rp[m];
      const double cross = rdp_inner4(D->res[m], old_pre[m]);
      D->rp[m] = rdp_inner4(D->res[m], D->pre[m]);
      const double beta = (D->rp[m] - cross) / before;
      if (D->since_restart[m] == RDP_CG_RESTART || beta <= 0 || before == 0) {
        memcpy(D->dir[m], D->pre[m], sizeof(D->dir[m]));
        D->since_restart[m] = 0;
      } else
        for (int i = 0; i < 4; i++) D->dir[m][i] = D->pre[m][i] + D->dir[m][i] * beta;
      D->since_restart[m]++;
    }
  }
}

/* a surviving candidate (four sides in w->out, their centre) -> rect_t */
static rect_t candidate_rect(const rdp_work *w, rdp_p2 centre, int iw, int ih, double tanAOV, uint32_t status) {
  rdp_rays R;
  descent2 D;
  int first;
  memset(&D, 0, sizeof(D));
  rdp_pose_setup(w->out, centre, iw, ih, tanAOV, &R, &first, D.t);
  descend(&R, &D);
  const double f[2] = { rdp_defect(&R, 0, D.t[0]), rdp_defect(&R, 1, D.t[1]) };
  rdp_rect r;
  rdp_pose_finish(w->out, first, &R, (const double (*)[4])D.t, f, status, &r);
  rect_t out;
  memcpy(&out, &r, sizeof(out));
  return out;
}

typedef struct { int *v; int n, cap; } intlist;
static void il_push(intlist *l, int x) {
  if (l->n == l->cap) { l->cap = l->cap ? l->cap * 2 : 16; l->v = (int *)realloc(l->v, sizeof(int) * (size_t)l->cap); }
  l->v[l->n++] = x;
}

/* ------------------------------------------------------------------ rh:1049-1226 */

void *rd_post_run(const void *segs, int max_records, const int *probes, int iw, int ih, double tanAOV) {
  const linesegment_t *ls = (const linesegment_t *)segs;
  int n = ((const int *)segs)[0];
  if (n > max_records - 1) n = max_records - 1;
  if (n < 0) n = 0;

  rect_t *ret = (rect_t *)calloc(16, sizeof(rect_t));
  int nret = 1, capret = 16;
#define PUSH_RECT(r) do { if (nret == capret) { capret *= 2; ret = (rect_t *)realloc(ret, sizeof(rect_t) * (size_t)capret); } ret[nret++] = (r); } while (0)
  /* work space of the funnel: a candidate never holds more segments than the frame has */
  rdp_work w;
  void *wmem = malloc(RDP_WORK_BYTES(n + 4));
  rdp_work_place(&w, wmem, n + 4);
  rdp_p2 centre;

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
    int na = 0;
    for (int j = 0; j < set->n; j++) {
      const int lsid = set->v[j];
      /* the voting-table slot of (lsid, segid): any probe of lsid that hit segid carries it */
      const int *e = NULL;
      for (int k = 0; k < 15; k++) { const int *pr = probes + (size_t)(lsid * 15 + k) * 6; if (pr[0] == segid) { e = pr + 1; break; } }
      rdp_seg whole;
      whole.e0 = rdp_pt(ls[lsid].x0, ls[lsid].y0); whole.e1 = rdp_pt(ls[lsid].x1, ls[lsid].y1);
      if (e[0] != lsid) {
        if (e[0] != 0) w.als[na++] = whole;
        continue;
      }
      double x0 = ls[lsid].x0, y0 = ls[lsid].y0, x1 = ls[lsid].x1, y1 = ls[lsid].y1;
      if (!rdp_clip(&x0, &y0, &x1, &y1, iw - e[1], ih - e[3], e[2], e[4])) continue;
      w.als[na].e0 = rdp_pt(x0, y0); w.als[na].e1 = rdp_pt(x1, y1);
      na++;
    }
    if (rdp_funnel(&w, na, &centre)) PUSH_RECT(candidate_rect(&w, centre, iw, ih, tanAOV, 0));
  }
  for (int gi = 0; gi < ngroups; gi++) { intlist *set = (intlist *)ArrayMap_get(groups, keys[gi]); free(set->v); free(set); }
  free(keys);
  ArrayMap_dispose(groups);

  /* pass 2: every polyline on its own, long segments only */
  for (int i = 1; i <= n; i++) {
    if (ls[i].polyid == 0 || ls[i].leftPtr > 0) continue;
    int na = 0;
    for (int j = i; j > 0 && j <= n && na < w.cap; j = ls[j].rightPtr) {
      const rdp_p2 e0 = rdp_pt(ls[j].x0, ls[j].y0), e1 = rdp_pt(ls[j].x1, ls[j].y1);
      if (rdp_d2(e0, e1) > 32.0 * 32.0) { w.als[na].e0 = e0; w.als[na].e1 = e1; na++; }
    }
    if (rdp_funnel(&w, na, &centre)) PUSH_RECT(candidate_rect(&w, centre, iw, ih, tanAOV, 2));
  }
  free(wmem);

  ret[0].nItems = nret;
  return ret;
}

/* Test tap: the probes of every segment taken on the host from full planes - what rh:1066-1098 reads from the planes it copied back -
 * then the post-process proper.  (The frame path takes the probes on the device: k_sample_segments, same rdp_probe_pixel.) */
void *rd_postprocess_planes(const void *segs, const int32_t *boundary, const int32_t *table, int iw, int ih, double tanAOV) {
  const linesegment_t *ls = (const linesegment_t *)segs;
  const int n = ((const int *)segs)[0];
  const unsigned nentry = (unsigned)(iw * ih * 4 / 5);
  int *probes = (int *)calloc((size_t)(n + 1) * 15 * 6, sizeof(int));
  for (int i = 1; i <= n; i++) {
    if (ls[i].polyid == 0) continue;
    for (int k = 0; k < 15; k++) {
      int sx, sy;
      if (!rdp_probe_pixel(ls[i].x0, ls[i].y0, ls[i].x1, ls[i].y1, k, iw, ih, &sx, &sy)) continue;
      int *rec = probes + (size_t)(i * 15 + k) * 6;
      const int bid = boundary[sx + sy * iw];
      rec[0] = bid;
      if (bid <= 0) continue;
      const int32_t *slot = table + (size_t)((((uint32_t)i * (uint32_t)bid) & 0x7fffffffu) % nentry) * 5;      /* the vote table's slot of (segment, boundary): rc:426-464 */
      for (int q = 0; q < 5; q++) rec[1 + q] = slot[q];
    }
  }
  void *r = rd_post_run(segs, n + 1, probes, iw, ih, tanAOV);
  free(probes);
  return r;
}


void rd_probe_pixels(float x0, float y0, float x1, float y1, int iw, int ih, int32_t *out) {
  for (int k = 0; k < 15; k++) {
    int sx, sy;
    if (rdp_probe_pixel(x0, y0, x1, y1, k, iw, ih, &sx, &sy)) { out[2 * k] = sx; out[2 * k + 1] = sy; }
    else { out[2 * k] = -1; out[2 * k + 1] = -1; }
  }
}
