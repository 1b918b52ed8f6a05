// rectdetect-mi355x: line segments + probes -> rectangles ON THE DEVICE (behaviour of the reference's executeCPUTask,
// oclrect.c:1049-1226, "rh").  Two launches per batch of frames (frame = blockIdx.z, see rdk::PolyFrame):
//
//   k_post_candidates  one block per frame.  Candidates in the reference's order: first the boundary components that at least four
//                      line segments run along (rh:1103-1131) - the reference walks a hash map whose iteration order is (bucket of the
//                      component id, order of first insertion), reproduced here by ranking - then every polyline chain (rh:1163-1188).
//   k_post_solve       one wave per candidate.  Lane 0 assembles the candidate's segments and runs the funnel (rd_post_core.h, the very
//                      source the host path compiles: hull, longest four sides, corners); the pose descent then uses the wave: the two
//                      side pairings are lane groups, the 9 residual evaluations of an axis stencil / the 3 of a line stencil sit on
//                      lanes and are exchanged by shuffles, every lane of a group carries the same descent state (computed redundantly:
//                      same inputs, same IEEE operations, same bits).
//
// Results go to pinned host memory: {count, flags}, one validity word and one rect_t per candidate, in candidate order; the host only
// packs the valid ones.  Anything that does not fit the fixed capacities (POST_* below) raises the overflow flag and the host path takes
// the frame.  Double precision, no contraction: the rectangles are bit-identical to the host path's (tests).
#include "rd_device.h"
#include "rd_kernels.h"
#include "rd_poly_scratch.h"
#include "rd_post_core.h"

namespace {

using rdk::PolyFrame;
using rdk::PolyFrames;

#define POST_HT 8192            // hash slots for component ids met by probes (power of two)
#define POST_MAXG 2048          // components that are candidates (>= 4 segments)
#define POST_MAXC RD_POST_MAXC  // candidates per frame
#define POST_MEMBERS 65536      // (segment, component) pairs of candidate components
#define POST_CAP 160            // segments per candidate in the funnel's work space
#define POST_WAVES 256          // waves (blocks of 64) per frame in k_post_solve

struct ls_rec { float x0, y0, x1, y1; int startIndex, endIndex, leftPtr, rightPtr, startCount, endCount, maxDist, polyid, npix, level; };

// layout of PolyFrame::post_scratch (ints)
#define PS_HT 0                                   // POST_HT x {key, first position, count, group index}
#define PS_GRP (PS_HT + POST_HT * 4)              // POST_MAXG x {key, first position, count, member offset}
#define PS_CAND (PS_GRP + POST_MAXG * 4)          // POST_MAXC x {type, key or first segment, member offset, member count}
#define PS_MEM (PS_CAND + POST_MAXC * 4)          // POST_MEMBERS
#define PS_FILL (PS_MEM + POST_MEMBERS)           // POST_MAXG fill counters
#define PS_CTR (PS_FILL + POST_MAXG)              // [0] groups, [1] candidates, [2] overflow, [3] members
#define PS_END (PS_CTR + 64)

__device__ __forceinline__ int am_bucket(unsigned k) { return (int)((k ^ (k >> 10) ^ (k >> 20) ^ (k >> 30)) & 1023u); }      // helper.c:127-134 on a 32-bit key
__device__ __forceinline__ unsigned ht_hash(int key) { return ((unsigned)key * 2654435761u) >> 19; }      // 13 bits

__device__ __forceinline__ int ht_find(const int *ht, int key) {
  unsigned h = ht_hash(key);
  for (int probes = 0; probes < POST_HT; probes++) {
    const int k = ht[h * 4];
    if (k == key) return (int)h;
    if (k == 0) return -1;
    h = (h + 1) & (POST_HT - 1);
  }
  return -1;
}

__global__ __launch_bounds__(1024) void k_post_candidates(const PolyFrames FRS, int max_records) {
  const PolyFrame &FRM = FRS.f[blockIdx.z];
  int *S = FRM.post_scratch;
  int *ht = S + PS_HT, *grp = S + PS_GRP, *cand = S + PS_CAND, *mem = S + PS_MEM, *fill = S + PS_FILL, *ctr = S + PS_CTR;
  const ls_rec *ls = (const ls_rec *)FRM.lslist;
  const int *probes = FRM.probes;
  const int tid = threadIdx.x;
  __shared__ int s_part[1024];
  int n = *(const int *)ls;
  if (n > max_records - 1) n = max_records - 1;
  if (n < 0) n = 0;
  for (int t = tid; t < POST_HT * 4; t += 1024) ht[t] = 0;
  for (int t = tid; t < POST_MAXG; t += 1024) fill[t] = 0;
  if (tid < 64) ctr[tid] = 0;
  __syncthreads();
  // (a) distinct component ids under the probes of valid segments: first position (segment * 15 + probe) and number of segments
  for (int t = tid; t < n * 15; t += 1024) {
    const int i = t / 15 + 1, k = t % 15;
    if (ls[i].polyid == 0) continue;
    const int *pr = probes + (size_t)(i * 15) * 6;
    const int key = pr[k * 6];
    if (key <= 0) continue;
    bool dup = false;
    for (int q = 0; q < k; q++) dup = dup || pr[q * 6] == key;
    if (dup) continue;                           // (a segment counts once per component: its first probe that hit it)
    unsigned h = ht_hash(key);
    int probes_n = 0;
    for (;;) {
      const int prev = atomicCAS(&ht[h * 4], 0, key);
      if (prev == 0 || prev == key) {
        if (prev == 0) ht[h * 4 + 1] = 0x7fffffff;
        break;
      }
      h = (h + 1) & (POST_HT - 1);
      if (++probes_n >= POST_HT) { ctr[2] = 1; break; }
    }
    if (probes_n < POST_HT) atomicAdd(&ht[h * 4 + 2], 1);
  }
  __syncthreads();
  // (first positions in a second sweep: the slot's position word is initialised by whoever claimed it, possibly after others arrived)
  for (int t = tid; t < POST_HT; t += 1024) if (ht[t * 4] != 0) ht[t * 4 + 1] = 0x7fffffff;
  __syncthreads();
  for (int t = tid; t < n * 15; t += 1024) {
    const int i = t / 15 + 1, k = t % 15;
    if (ls[i].polyid == 0) continue;
    const int key = probes[(size_t)(i * 15 + k) * 6];
    if (key <= 0) continue;
    const int h = ht_find(ht, key);
    if (h >= 0) atomicMin(&ht[h * 4 + 1], i * 15 + k);
  }
  __syncthreads();
  // (b) the components with at least four segments, in any order
  for (int t = tid; t < POST_HT; t += 1024) {
    ht[t * 4 + 3] = -1;
    if (ht[t * 4] != 0 && ht[t * 4 + 2] >= 4) {
      const int g = atomicAdd(&ctr[0], 1);
      if (g < POST_MAXG) { grp[g * 4] = ht[t * 4]; grp[g * 4 + 1] = ht[t * 4 + 1]; grp[g * 4 + 2] = ht[t * 4 + 2]; grp[g * 4 + 3] = t; }
      else ctr[2] = 1;
    }
  }
  __syncthreads();
  const int ng = min(ctr[0], POST_MAXG);
  // (c) their rank in the reference's order: (bucket of the id, position of first insertion)
  for (int g = tid; g < ng; g += 1024) {
    const int key = grp[g * 4], pos = grp[g * 4 + 1];
    const int b = am_bucket((unsigned)key);
    int rank = 0;
    for (int o = 0; o < ng; o++) {
      const int ob = am_bucket((unsigned)grp[o * 4]), op = grp[o * 4 + 1];
      rank += (ob < b || (ob == b && op < pos)) ? 1 : 0;
    }
    if (rank < POST_MAXC) { cand[rank * 4] = 0; cand[rank * 4 + 1] = key; cand[rank * 4 + 3] = grp[g * 4 + 2]; ht[grp[g * 4 + 3] * 4 + 3] = rank; }
    else ctr[2] = 1;
  }
  __syncthreads();
  if (tid == 0) {          // member offsets in rank order
    int off = 0;
    const int m = min(ng, POST_MAXC);
    for (int c = 0; c < m; c++) { cand[c * 4 + 2] = off; off += cand[c * 4 + 3]; }
    ctr[3] = off;
    if (off > POST_MEMBERS) ctr[2] = 1;
  }
  __syncthreads();
  // (d) the segments of every candidate component (any order here; the solver sorts each short list)
  if (ctr[2] == 0)
    for (int t = tid; t < n * 15; t += 1024) {
      const int i = t / 15 + 1, k = t % 15;
      if (ls[i].polyid == 0) continue;
      const int *pr = probes + (size_t)(i * 15) * 6;
      const int key = pr[k * 6];
      if (key <= 0) continue;
      bool dup = false;
      for (int q = 0; q < k; q++) dup = dup || pr[q * 6] == key;
      if (dup) continue;
      const int h = ht_find(ht, key);
      const int c = h >= 0 ? ht[h * 4 + 3] : -1;
      if (c < 0) continue;
      mem[cand[c * 4 + 2] + atomicAdd(&fill[c], 1)] = i;
    }
  // (e) chains: the segments without a left neighbour, in index order, appended behind the components
  const int per = (n + 1023) / 1024;
  int cnt = 0;
  for (int q = 0; q < per; q++) { const int i = 1 + tid * per + q; if (i <= n && ls[i].polyid != 0 && !(ls[i].leftPtr > 0)) cnt++; }
  s_part[tid] = cnt;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = tid >= o ? s_part[tid - o] : 0;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  int at = min(ng, POST_MAXC) + s_part[tid] - cnt;
  for (int q = 0; q < per; q++) {
    const int i = 1 + tid * per + q;
    if (i <= n && ls[i].polyid != 0 && !(ls[i].leftPtr > 0)) {
      if (at < POST_MAXC) { cand[at * 4] = 1; cand[at * 4 + 1] = i; cand[at * 4 + 2] = 0; cand[at * 4 + 3] = 0; } else ctr[2] = 1;
      at++;
    }
  }
  if (tid == 1023) ctr[1] = min(min(ng, POST_MAXC) + s_part[1023], POST_MAXC);
  // the header of the result block is written by the solver's first wave once it knows about overflows of its own
}

// gathers of a double from lane `src` (all lanes of the wave take part)
__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src); }

__global__ __launch_bounds__(64) void k_post_solve(const PolyFrames FRS, int max_records, int iw, int ih, double tanAOV) {
  const PolyFrame &FRM = FRS.f[blockIdx.z];
  int *S = FRM.post_scratch;
  const int *cand = S + PS_CAND, *mem = S + PS_MEM, *ctr = S + PS_CTR;
  const ls_rec *ls = (const ls_rec *)FRM.lslist;
  const int *probes = FRM.probes;
  int *outb = FRM.post_out;                      // [0] candidates, [1] overflow, [2] tan (2 ints), [8 + c] validity, rects behind
  rdp_rect *rects = (rdp_rect *)(outb + 8 + POST_MAXC);
  const int lane = threadIdx.x;
  const int ncand = ctr[1];
  int n = *(const int *)ls;
  if (n > max_records - 1) n = max_records - 1;
  __shared__ rdp_rays R;
  __shared__ double t0[2][4];
  __shared__ int s_ok, s_first, s_over;
  __shared__ rdp_p2 s_centre;
  // the funnel's work space (35 KB) lives in LDS: lane 0 walks through it one access after the other, a trip to global memory each
  // time made the kernel three times as long
  __shared__ __align__(16) unsigned char wmem[RDP_WORK_BYTES(POST_CAP)];
  rdp_work w;
  rdp_work_place(&w, wmem, POST_CAP);
  if (lane == 0) s_over = 0;
  for (int c = blockIdx.x; c < ncand; c += gridDim.x) {
    const int type = cand[c * 4], key = cand[c * 4 + 1];
    if (lane == 0) {
      // the candidate's segments (rh:1109-1131 / rh:1168-1176)
      int na = 0, ok = 0;
      w.overflow = 0;
      if (type == 0) {
        const int off = cand[c * 4 + 2], cnt = cand[c * 4 + 3];
        int last = 0;
        for (int j = 0; j < cnt && !w.overflow; j++) {
          int lsid = 0x7fffffff;                   // members in ascending order: the smallest one above the previous
          for (int q = 0; q < cnt; q++) { const int v = mem[off + q]; if (v > last && v < lsid) lsid = v; }
          last = lsid;
          const int *e = nullptr;
          for (int k = 0; k < 15; k++) { const int *pr = probes + (size_t)(lsid * 15 + k) * 6; if (pr[0] == key) { e = pr + 1; break; } }
          rdp_seg whole;
          whole.e0 = rdp_pt(ls[lsid].x0, ls[lsid].y0); whole.e1 = rdp_pt(ls[lsid].x1, ls[lsid].y1);
          if (na >= POST_CAP) { w.overflow = 1; break; }
          if (e[0] != lsid) { if (e[0] != 0) w.als[na++] = whole; continue; }
          double x0 = ls[lsid].x0, y0 = ls[lsid].y0, x1 = ls[lsid].x1, y1 = ls[lsid].y1;
          if (!rdp_clip(&x0, &y0, &x1, &y1, iw - e[1], ih - e[3], e[2], e[4])) continue;
          w.als[na].e0 = rdp_pt(x0, y0); w.als[na].e1 = rdp_pt(x1, y1);
          na++;
        }
      } else {
        for (int j = key; j > 0 && j <= n; j = ls[j].rightPtr) {
          const rdp_p2 e0 = rdp_pt(ls[j].x0, ls[j].y0), e1 = rdp_pt(ls[j].x1, ls[j].y1);
          if (rdp_d2(e0, e1) > 32.0 * 32.0) { if (na >= POST_CAP) { w.overflow = 1; break; } w.als[na].e0 = e0; w.als[na].e1 = e1; na++; }
        }
      }
      rdp_p2 centre = rdp_pt(0, 0);
      if (!w.overflow) ok = rdp_funnel(&w, na, &centre);
      if (w.overflow) { s_over = 1; ok = 0; }
      if (ok) {
        int first;
        rdp_pose_setup(w.out, centre, iw, ih, tanAOV, &R, &first, t0);
        s_first = first;
      }
      s_ok = ok; s_centre = centre;
    }
    __syncthreads();
    if (!s_ok) { if (lane == 0) outb[8 + c] = 0; __syncthreads(); continue; }

    // ---- the descent (rh:479-588), both pairings at once: lanes 0-8 pairing 0, lanes 9-17 pairing 1, the rest mirror lane 0's group
    const int m = (lane >= 9 && lane < 18) ? 1 : 0, j = lane < 18 ? lane % 9 : 0, g0 = m * 9;
    double t[4], res[4], pre[4], dir[4], rp = 0;
    int since = 0;
    for (int i = 0; i < 4; i++) t[i] = t0[m][i];
    // central differences along the four axes: lane j evaluates the residual at the centre (0), one step down (1..4) or up (5..8) an axis
    auto probe = [&]() {
      double pt[4];
      for (int i = 0; i < 4; i++) { const double h = ((j >= 1 && j <= 4 && j - 1 == i) || (j >= 5 && j - 5 == i)) ? RDP_FD_STEP : 0; pt[i] = (j >= 5) ? t[i] + h : t[i] - h; }
      const double fv = rdp_defect(&R, m, pt);
      const double f0 = shfl_d(fv, g0);
      double curv[4];
      bool convex = true;
      for (int i = 0; i < 4; i++) {
        const double fl = shfl_d(fv, g0 + 1 + i), fh = shfl_d(fv, g0 + 5 + i);
        res[i] = (fh - fl) / (2 * RDP_FD_STEP) * -1;
        curv[i] = (fl - 2 * f0 + fh) / (RDP_FD_STEP * RDP_FD_STEP);
        convex = convex && !(curv[i] <= 0);
      }
      for (int i = 0; i < 4; i++) {
        if (convex) { pre[i] = 1.0 / curv[i]; pre[i] *= res[i]; } else pre[i] = res[i];
      }
    };
    probe();
    for (int i = 0; i < 4; i++) dir[i] = pre[i];
    rp = rdp_inner4(res, dir);
    for (int it = 0; it < RDP_CG_STEPS; it++) {
      {   // damped Newton steps along dir: lanes j % 3 = 0 / 1 / 2 evaluate the residual at the point, one step ahead, one step back
        const double k = 1.0 / (sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2] + dir[3] * dir[3]) + 1e-20);
        double u[4], damp = 1.0;
        bool done = false;
        for (int i = 0; i < 4; i++) u[i] = dir[i] * k;
        for (int step = 0; step < RDP_WALK_STEPS; step++) {
          double pt[4], cnd[4];
          const int role = j % 3;
          for (int i = 0; i < 4; i++) pt[i] = role == 0 ? t[i] : (role == 1 ? t[i] + u[i] * RDP_FD_STEP : t[i] + u[i] * -RDP_FD_STEP);
          const double fv = rdp_defect(&R, m, pt);
          const double f0 = shfl_d(fv, g0), ff = shfl_d(fv, g0 + 1), fb = shfl_d(fv, g0 + 2);
          const double slope = (ff - fb) * (1.0 / (2 * RDP_FD_STEP));
          double curv = (ff + fb - 2 * f0) * (1.0 / (RDP_FD_STEP * RDP_FD_STEP));
          if (curv * curv < 1e-10) curv = 1;
          const double len = fabs(slope / curv);
          const bool stop = len < 1e-10;
          for (int i = 0; i < 4; i++) cnd[i] = t[i] + u[i] * (len * damp);
          const double e1 = rdp_defect(&R, m, cnd);
          if (!done && !stop) {
            if (f0 < e1) damp *= 0.5;
            else for (int i = 0; i < 4; i++) t[i] = cnd[i];
          }
          done = done || stop;
        }
      }
      double old_pre[4];
      for (int i = 0; i < 4; i++) old_pre[i] = pre[i];
      probe();
      const double before = rp;
      const double cross = rdp_inner4(res, old_pre);
      rp = rdp_inner4(res, pre);
      const double beta = (rp - cross) / before;
      if (since == RDP_CG_RESTART || beta <= 0 || before == 0) { for (int i = 0; i < 4; i++) dir[i] = pre[i]; since = 0; }
      else for (int i = 0; i < 4; i++) dir[i] = pre[i] + dir[i] * beta;
      since++;
    }
    const double fmine = rdp_defect(&R, m, t);
    double tt[2][4], ff2[2];
    for (int q = 0; q < 2; q++) { ff2[q] = shfl_d(fmine, q * 9); for (int i = 0; i < 4; i++) tt[q][i] = shfl_d(t[i], q * 9); }
    if (lane == 0) {
      rdp_rect r;
      rdp_pose_finish(w.out, s_first, &R, (const double (*)[4])tt, ff2, type == 0 ? 0u : 2u, &r);
      rects[c] = r;
      outb[8 + c] = 1;
    }
    __syncthreads();
  }
  __syncthreads();
  if (lane == 0 && s_over) atomicExch(&S[PS_CTR + 2], 1);
}

// the header of the result block: after all solver waves (stream order)
__global__ void k_post_header(const PolyFrames FRS, double tanAOV) {
  const PolyFrame &FRM = FRS.f[blockIdx.z];
  const int *ctr = FRM.post_scratch + PS_CTR;
  int *outb = FRM.post_out;
  if (threadIdx.x == 0) {
    outb[1] = ctr[2];
    *(double *)(outb + 2) = tanAOV;
    __threadfence_system();
    outb[0] = ctr[1];
  }
}

}  // namespace

namespace rdk {

size_t post_scratch_ints() { return (size_t)PS_END; }
size_t post_out_ints() { return 8 + (size_t)POST_MAXC + (size_t)POST_MAXC * (sizeof(rdp_rect) / sizeof(int)); }

// rectangles of nb frames from their segment lists and probes (frames[z].probes / lslist as sample_segments left them) into frames[z].post_out
void post_device(hipStream_t st, const PolyFrame *frames_host, int nb, int max_records, int iw, int ih, double tanAOV) {
  const PolyFrames frames = pack_frames(frames_host, nb);
  hipLaunchKernelGGL(k_post_candidates, dim3(1, 1, nb), dim3(1024), 0, st, frames, max_records);
  hipLaunchKernelGGL(k_post_solve, dim3(POST_WAVES, 1, nb), dim3(64), 0, st, frames, max_records, iw, ih, tanAOV);
  hipLaunchKernelGGL(k_post_header, dim3(1, 1, nb), dim3(64), 0, st, frames, tanAOV);
}

}  // namespace rdk
