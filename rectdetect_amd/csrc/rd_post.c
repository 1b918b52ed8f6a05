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
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

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
  double f0[RDP_PAIRINGS];          /* residual at t, when have_f0 */
  int have_f0[RDP_PAIRINGS];
} descent2;

/* central differences along the four axes (rh:492-512) -> res = -gradient; pre = res / curvature per axis when every
 * curvature is positive, else res (rh:538-555).  D->f0[m] = the residual at the current depths (the reference evaluates it again here:
 * the same arguments, the same value - taken from the walk that produced the depths when there was one) */
static void probe_axes(const rdp_rays *R, descent2 *D) {
  for (int m = 0; m < RDP_PAIRINGS; m++) {
    if (!D->have_f0[m]) { D->f0[m] = rdp_defect(R, m, D->t[m]); D->have_f0[m] = 1; }
    const double f0 = D->f0[m];
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
 * whenever it does not lower the residual, a pairing stops once its step falls below 1e-10.  The reference evaluates the residual four
 * times per step (centre, both stencil points, the candidate); a step that was refused leaves the depths - hence the first three and
 * everything derived from them - as they were, and an accepted candidate's residual is the next step's centre: those evaluations are
 * reused instead of repeated (the same arguments give the same bits), which leaves three per accepted and one per refused step. */
static void walk_along(const rdp_rays *R, descent2 *D) {
  for (int m = 0; m < RDP_PAIRINGS; m++) {
    double u[4], damp = 1.0, len = 0;
    const double k = 1.0 / (sqrt(D->dir[m][0] * D->dir[m][0] + D->dir[m][1] * D->dir[m][1] + D->dir[m][2] * D->dir[m][2] + D->dir[m][3] * D->dir[m][3]) + 1e-20);
    for (int i = 0; i < 4; i++) u[i] = D->dir[m][i] * k;
    if (!D->have_f0[m]) { D->f0[m] = rdp_defect(R, m, D->t[m]); D->have_f0[m] = 1; }
    int have_len = 0;
    for (int step = 0; step < RDP_WALK_STEPS; step++) {
      double cand[4];
      if (!have_len) {
        double fwd[4], bwd[4];
        for (int i = 0; i < 4; i++) { fwd[i] = D->t[m][i] + u[i] * RDP_FD_STEP; bwd[i] = D->t[m][i] + u[i] * -RDP_FD_STEP; }
        const double f0 = D->f0[m], ff = rdp_defect(R, m, fwd), fb = rdp_defect(R, m, bwd);
        const double slope = (ff - fb) * (1.0 / (2 * RDP_FD_STEP));
        double curv = (ff + fb - 2 * f0) * (1.0 / (RDP_FD_STEP * RDP_FD_STEP));
        if (curv * curv < 1e-10) curv = 1;
        len = fabs(slope / curv);
        have_len = 1;
      }
      if (len < 1e-10) break;
      for (int i = 0; i < 4; i++) cand[i] = D->t[m][i] + u[i] * (len * damp);
      const double fc = rdp_defect(R, m, cand);
      if (D->f0[m] < fc) { damp *= 0.5; continue; }
      for (int i = 0; i < 4; i++) D->t[m][i] = cand[i];
      D->f0[m] = fc; have_len = 0;
    }
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

/* a surviving candidate (its four sides in angular order, their centre) -> rect_t */
static rect_t candidate_rect(const rdp_seg *sides, rdp_p2 centre, int iw, int ih, double tanAOV, uint32_t status) {
  rdp_rays R;
  descent2 D;
  int first;
  memset(&D, 0, sizeof(D));
  rdp_pose_setup(sides, centre, iw, ih, tanAOV, &R, &first, D.t);
  descend(&R, &D);
  const double f[2] = { D.have_f0[0] ? D.f0[0] : rdp_defect(&R, 0, D.t[0]), D.have_f0[1] ? D.f0[1] : rdp_defect(&R, 1, D.t[1]) };
  rdp_rect r;
  rdp_pose_finish(sides, first, &R, (const double (*)[4])D.t, f, status, &r);
  rect_t out;
  memcpy(&out, &r, sizeof(out));
  return out;
}

/* ------------------------------------------------------------------ the pose estimations of a frame's candidates, side by side
 * The funnel leaves ~a dozen candidates per 1080p frame and the descent of one takes 20-25 us on a core: with the caller's thread alone that is
 * a quarter of a millisecond at the end of EVERY frame's latency (the reference does the same on its caller's thread, rh:1049-1226).  The
 * candidates are independent, so helper threads take some of them - but a sleeping thread takes longer to wake than its share is worth.  So
 * the caller ARMS the helpers when it begins to wait for the device (rd_post_helpers_arm): they wake while the device is still busy and spin,
 * for a bounded time, until the candidates are published; then everybody - the caller included - claims candidates one by one.  Results
 * land in candidate order: the returned list does not depend on who computed what.  One frame at a time: a caller that finds the helpers
 * taken (another detector's poll) runs its candidates alone.
 * The helpers live as long as the process (never joined; a child of fork() has none and runs everything on its caller's thread: the caller claims every
 * job nobody else claims, so nothing ever waits for a helper that does not exist; the library is not meant to be unloaded while they exist). */
typedef struct { rdp_seg sides[4]; rdp_p2 centre; uint32_t status; rect_t out; } pose_job;
typedef struct { pose_job *jobs; int iw, ih; double tan_aov; } pose_batch;

#if defined(__x86_64__) || defined(__i386__)
#define RD_CPU_RELAX() __builtin_ia32_pause()
#else
#define RD_CPU_RELAX() __asm__ __volatile__("" ::: "memory")
#endif

static struct {
  pthread_mutex_t mu;             /* guards everything below (the three words the spinning helpers read without it are written with atomic stores) */
  pthread_cond_t cv;
  pthread_mutex_t owner;          /* the caller whose jobs the helpers work on */
  pthread_t th[RD_POST_MAX_HELPERS];
  int nthreads, quit;             /* nthreads: atomic (read by callers that do not hold the lock) */
  unsigned arm_gen;               /* raised by every rd_post_helpers_arm; atomic */
  rd_job_fn fn; void *ctx; int njobs, next, done;
  int pending;                    /* njobs - next; atomic */
  int atfork_registered;
} pool = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_MUTEX_INITIALIZER };

static double mono_us(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }

/* claims and runs jobs (in index order) until none is left; returns how many this thread ran */
static int pool_work(rd_progress_fn progress) {
  int ran = 0;
  for (;;) {
    pthread_mutex_lock(&pool.mu);
    if (pool.next >= pool.njobs) { pthread_mutex_unlock(&pool.mu); return ran; }
    const int idx = pool.next++;
    __atomic_store_n(&pool.pending, pool.njobs - pool.next, __ATOMIC_RELAXED);
    const rd_job_fn fn = pool.fn; void *ctx = pool.ctx;
    pthread_mutex_unlock(&pool.mu);
    fn(ctx, idx);
    ran++;
    pthread_mutex_lock(&pool.mu);
    pool.done++;
    pthread_mutex_unlock(&pool.mu);
    if (progress) progress(ctx);
  }
}

static void *pool_helper(void *arg) {
  (void)arg;
  unsigned seen = 0;
  for (;;) {
    pthread_mutex_lock(&pool.mu);
    while (__atomic_load_n(&pool.arm_gen, __ATOMIC_RELAXED) == seen && !pool.quit) pthread_cond_wait(&pool.cv, &pool.mu);
    seen = __atomic_load_n(&pool.arm_gen, __ATOMIC_RELAXED);
    if (pool.quit) { pthread_mutex_unlock(&pool.mu); return NULL; }
    pthread_mutex_unlock(&pool.mu);
    /* armed: a caller has work coming - spin until jobs appear, take part, and go on (the same call brings the copy of a frame first and its candidates a
     * millisecond later) until RD_POST_SPIN_US have passed since the last arm: a frame that takes longer than that is not one whose latency these threads can
     * save.  A call of arm that arrives meanwhile extends the time.  The first RD_POST_HOT_US are a tight spin; after that the thread offers its core to
     * whoever wants it between two looks (sched_yield): on a busy machine an armed helper costs next to nothing, on an idle one it is what it is meant to be. */
    const double armed = mono_us();
    double until = armed + RD_POST_SPIN_US, hot = armed + RD_POST_HOT_US;
    for (;;) {
      if (__atomic_load_n(&pool.pending, __ATOMIC_RELAXED) > 0) { pool_work(NULL); hot = mono_us() + RD_POST_HOT_US; }
      else if (__atomic_load_n(&pool.arm_gen, __ATOMIC_RELAXED) != seen) { seen = __atomic_load_n(&pool.arm_gen, __ATOMIC_RELAXED); until = mono_us() + RD_POST_SPIN_US; }      /* (a change missed here is caught by the wait above) */
      else {
        const double now = mono_us();
        if (now > until || __atomic_load_n(&pool.quit, __ATOMIC_RELAXED)) break;
        if (now > hot) sched_yield(); else RD_CPU_RELAX();
      }
    }
  }
}

/* the child of a fork() has none of the parent's threads: it starts with an empty pool (and fresh locks - a helper may have held one at the fork) and runs
 * everything on its caller's thread until it configures helpers of its own */
static void pool_after_fork_in_child(void) {
  pthread_mutex_init(&pool.mu, NULL); pthread_cond_init(&pool.cv, NULL); pthread_mutex_init(&pool.owner, NULL);
  __atomic_store_n(&pool.nthreads, 0, __ATOMIC_RELAXED);
  pool.quit = 0; pool.fn = NULL; pool.ctx = NULL; pool.njobs = pool.next = pool.done = 0;
  __atomic_store_n(&pool.pending, 0, __ATOMIC_RELAXED);
}

void rd_post_helpers_configure(int n) {
  if (n > RD_POST_MAX_HELPERS) n = RD_POST_MAX_HELPERS;
  pthread_mutex_lock(&pool.mu);
  if (!pool.atfork_registered) { pthread_atfork(NULL, NULL, pool_after_fork_in_child); pool.atfork_registered = 1; }
  while (pool.nthreads < n) {
    if (pthread_create(&pool.th[pool.nthreads], NULL, pool_helper, NULL) != 0) break;
    __atomic_store_n(&pool.nthreads, pool.nthreads + 1, __ATOMIC_RELEASE);
  }
  pthread_mutex_unlock(&pool.mu);
}

/* the helpers end with the library (dlclose, exit): asked to quit and joined, so that none of them is left running code that is about to be unmapped */
void rd_post_helpers_shutdown(void) {
  pthread_mutex_lock(&pool.owner);      /* (no frame's jobs in flight) */
  pthread_mutex_lock(&pool.mu);
  const int n = pool.nthreads;
  __atomic_store_n(&pool.quit, 1, __ATOMIC_RELAXED);
  pthread_cond_broadcast(&pool.cv);
  pthread_mutex_unlock(&pool.mu);
  for (int i = 0; i < n; i++) pthread_join(pool.th[i], NULL);
  pthread_mutex_lock(&pool.mu);
  __atomic_store_n(&pool.nthreads, 0, __ATOMIC_RELEASE);
  pool.quit = 0;
  pthread_mutex_unlock(&pool.mu);
  pthread_mutex_unlock(&pool.owner);
}
__attribute__((destructor)) static void pool_at_unload(void) { if (__atomic_load_n(&pool.nthreads, __ATOMIC_ACQUIRE) > 0) rd_post_helpers_shutdown(); }

void rd_post_helpers_arm(void) {
  if (__atomic_load_n(&pool.nthreads, __ATOMIC_ACQUIRE) == 0) return;
  pthread_mutex_lock(&pool.mu);
  __atomic_store_n(&pool.arm_gen, pool.arm_gen + 1, __ATOMIC_RELAXED);
  pthread_cond_broadcast(&pool.cv);
  pthread_mutex_unlock(&pool.mu);
}

int rd_post_helpers(void) { return __atomic_load_n(&pool.nthreads, __ATOMIC_ACQUIRE); }

void rd_helpers_run(rd_job_fn fn, void *ctx, int n, rd_progress_fn progress) {
  if (n > 1 && __atomic_load_n(&pool.nthreads, __ATOMIC_ACQUIRE) > 0 && pthread_mutex_trylock(&pool.owner) == 0) {
    pthread_mutex_lock(&pool.mu);
    pool.fn = fn; pool.ctx = ctx; pool.njobs = n; pool.next = 0; pool.done = 0;
    __atomic_store_n(&pool.pending, n, __ATOMIC_RELAXED);
    pthread_mutex_unlock(&pool.mu);
    pool_work(progress);
    for (;;) {      /* jobs the helpers claimed: they are running them right now */
      pthread_mutex_lock(&pool.mu);
      const int fin = pool.done == pool.njobs;
      if (fin) { pool.njobs = 0; pool.next = 0; __atomic_store_n(&pool.pending, 0, __ATOMIC_RELAXED); pool.fn = NULL; pool.ctx = NULL; }
      pthread_mutex_unlock(&pool.mu);
      if (fin) break;
      if (progress) progress(ctx); else RD_CPU_RELAX();
    }
    pthread_mutex_unlock(&pool.owner);
    if (progress) progress(ctx);
    return;
  }
  for (int i = 0; i < n; i++) { fn(ctx, i); if (progress) progress(ctx); }
}

static void pose_job_run(void *ctx, int i) {
  pose_batch *b = (pose_batch *)ctx;
  b->jobs[i].out = candidate_rect(b->jobs[i].sides, b->jobs[i].centre, b->iw, b->ih, b->tan_aov, b->jobs[i].status);
}

static void run_pose_jobs(pose_job *jobs, int n, int iw, int ih, double tanAOV) {
  pose_batch b = { jobs, iw, ih, tanAOV };
  rd_helpers_run(pose_job_run, &b, n, NULL);
}

typedef struct { int *v; int n, cap; } intlist;
static void il_push(intlist *l, int x) {
  if (l->n == l->cap) { l->cap = l->cap ? l->cap * 2 : 16; l->v = (int *)realloc(l->v, sizeof(int) * (size_t)l->cap); }
  l->v[l->n++] = x;
}

/* ------------------------------------------------------------------ the frame's way into pinned memory */
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2"))) static void copy_stream_avx2(char *d, const char *s, size_t n) {
  while (((uintptr_t)d & 31) && n) { *d++ = *s++; n--; }
  for (; n >= 128; n -= 128, d += 128, s += 128) {
    const __m256i a = _mm256_loadu_si256((const __m256i *)s), b = _mm256_loadu_si256((const __m256i *)(s + 32));
    const __m256i c = _mm256_loadu_si256((const __m256i *)(s + 64)), e = _mm256_loadu_si256((const __m256i *)(s + 96));
    _mm256_stream_si256((__m256i *)d, a); _mm256_stream_si256((__m256i *)(d + 32), b);
    _mm256_stream_si256((__m256i *)(d + 64), c); _mm256_stream_si256((__m256i *)(d + 96), e);
  }
  _mm_sfence();
  if (n) memcpy(d, s, n);
}

void rd_copy_to_staging(void *dst, const void *src, size_t n) {
  static int avx2 = -1;
  int have = __atomic_load_n(&avx2, __ATOMIC_RELAXED);
  if (have < 0) { have = __builtin_cpu_supports("avx2") ? 1 : 0; __atomic_store_n(&avx2, have, __ATOMIC_RELAXED); }
  if (have && n >= 4096) copy_stream_avx2((char *)dst, (const char *)src, n);
  else memcpy(dst, src, n);
}
#else
void rd_copy_to_staging(void *dst, const void *src, size_t n) { memcpy(dst, src, n); }
#endif

/* ------------------------------------------------------------------ rh:1049-1226 */

void *rd_post_run(const void *segs, int max_records, const int *probes, int iw, int ih, double tanAOV) {
  const linesegment_t *ls = (const linesegment_t *)segs;
  int n = ((const int *)segs)[0];
  if (n > max_records - 1) n = max_records - 1;
  if (n < 0) n = 0;

  /* the candidates that survive the funnel, in the reference's list order; their poses are estimated afterwards, side by side */
  pose_job *jobs = (pose_job *)malloc(16 * sizeof(pose_job));
  int njobs = 0, capjobs = 16;
#define PUSH_CANDIDATE(st) do { if (njobs == capjobs) { capjobs *= 2; jobs = (pose_job *)realloc(jobs, sizeof(pose_job) * (size_t)capjobs); } \
    memcpy(jobs[njobs].sides, w.out, sizeof(jobs[njobs].sides)); jobs[njobs].centre = centre; jobs[njobs].status = (st); njobs++; } while (0)
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
    if (rdp_funnel(&w, na, &centre)) PUSH_CANDIDATE(0);
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
    if (rdp_funnel(&w, na, &centre)) PUSH_CANDIDATE(2);
  }
  free(wmem);

  run_pose_jobs(jobs, njobs, iw, ih, tanAOV);
  rect_t *ret = (rect_t *)calloc((size_t)njobs + 1, sizeof(rect_t));
  for (int i = 0; i < njobs; i++) ret[i + 1] = jobs[i].out;
  free(jobs);
  ret[0].nItems = njobs + 1;
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
