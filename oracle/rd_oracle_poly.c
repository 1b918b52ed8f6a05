/*
 * TEST INFRASTRUCTURE ONLY - CPU restatement of the polyline stage, oclpolyline_execute
 * (oclpolyline.c:218-309) and its kernels (oclpolyline.cl = "pl").  See rd_oracle.c for the rules.
 *
 * Order-dependent spots are resolved to serial raster order (H3, H7, H8, H15); the two
 * labelling steps are run to convergence (H4).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "rd_oracle.h"

static const int RX[8] = { 1, 1, 0, -1, -1, -1, 0, 1 }, RY[8] = { 0, -1, -1, -1, 0, 1, 1, 1 };

static inline int interior(int x, int y, int iw, int ih) { return x > 0 && y > 0 && x < iw - 1 && y < ih - 1; }

/* pl:89-110.  Ring (2 px) is not written by the kernel: it keeps whatever the scratch plane held. */
static void connect_poly(int *out, const int *in, int ring_value, int iw, int ih) {
  for (int y = 0; y < ih; y++)
    for (int x = 0; x < iw; x++) {
      const int p = y * iw + x;
      if (x <= 1 || y <= 1 || x >= iw - 2 || y >= ih - 2) { out[p] = ring_value; continue; }
      if (in[p] != 0) { out[p] = 1; continue; }
      int o = 0;
      if (in[p - 2] != 0 && in[p - 1] == 2 && in[p + 1] == 2 && in[p + 2] != 0) o = 1;
      if (in[p - iw * 2] != 0 && in[p - iw] == 2 && in[p + iw] == 2 && in[p + iw * 2] != 0) o = 1;
      if (in[p - iw * 2 - 2] != 0 && in[p - iw - 1] == 2 && in[p + iw + 1] == 2 && in[p + iw * 2 + 2] != 0) o = 1;
      if (in[p - iw * 2 + 2] != 0 && in[p - iw + 1] == 2 && in[p + iw - 1] == 2 && in[p + iw * 2 - 2] != 0) o = 1;
      if (in[p + 2] != 0 && in[p + 1] == 2 && in[p + iw - 1] == 2 && in[p + iw - 2] != 0) o = 1;
      if (in[p - 2] != 0 && in[p - 1] == 2 && in[p + iw + 1] == 2 && in[p + iw + 2] != 0) o = 1;
      if (in[p - iw * 2 + 1] != 0 && in[p - iw + 1] == 2 && in[p + iw] == 2 && in[p + iw * 2] != 0) o = 1;
      if (in[p - iw * 2 - 1] != 0 && in[p - iw - 1] == 2 && in[p + iw] == 2 && in[p + iw * 2] != 0) o = 1;
      out[p] = o;
    }
}

/* pl:126-147 */
static void remove_branch(int *out, const int *in, int iw, int ih) {
  memset(out, 0, sizeof(int) * (size_t)iw * ih);
  for (int y = 1; y < ih - 1; y++)
    for (int x = 1; x < iw - 1; x++) {
      const int p = y * iw + x;
      if (in[p] == 0) continue;
      int count = 0;
      for (int i = 0; i < 8; i++) if (in[p + RX[i] + RY[i] * iw] != 0) count++;
      out[p] = count <= 2 ? 1 : 0;
    }
}

/* pl:169-191: the first two 8-neighbours (E, NE, N, NW, W, SW, S, SE order) carrying the same label */
static void two_neighbours(const int *label, int p0, int iw, int *a, int *b) {
  const int l = label[p0];
  int i;
  for (i = 0; i < 8; i++) if (label[p0 + RX[i] + RY[i] * iw] == l) break;
  *a = i < 8 ? p0 + RX[i] + RY[i] * iw : p0;
  for (i++; i < 8; i++) if (label[p0 + RX[i] + RY[i] * iw] == l) break;
  *b = i < 8 ? p0 + RX[i] + RY[i] * iw : p0;
}

/* pl:870-889: per-pixel 64-bit hash (seed 0, oclpolyline.c:199) */
static uint64_t rotl64(uint64_t t, int n) { n &= 63; return n ? (t << n) | (t >> (64 - n)) : t; }
static uint64_t mix64(uint64_t s) {
  static const uint64_t K[7] = { 0xf3dd0fb7820fde37ULL, 0xe6c6ac2c59e52811ULL, 0x2fc7871fff7c5b45ULL, 0x47c7e1f70aa4f7c5ULL,
                                 0x094f02b7fb9ba895ULL, 0x89afda817e744570ULL, 0xc7277d052c7bf14bULL };
  static const int SH[7] = { 24, 6, 18, 48, 0, 12, 36 };
  uint64_t t = s;
  for (int i = 0; i < 7; i++) { t = rotl64(t, (int)((s >> SH[i]) & 63)); t ^= K[i]; }
  return t;
}
static inline int pixel_rand(int p, uint64_t seed) {
  return (int)mix64(((uint64_t)(int64_t)p ^ 0xb21c2cb635b48285ULL) * 0x9b923b9cec745401ULL + (seed ^ 0x7bb93d75a79d2f15ULL) * 0x22cab58ada573a29ULL);
}

static inline float dist2f(float vx, float vy, float wx, float wy) { return (vx - wx) * (vx - wx) + (vy - wy) * (vy - wy); }

/* pl:51-59 */
static void closest_on_segment(float vx, float vy, float wx, float wy, float px, float py, float *cx, float *cy) {
  float l2 = dist2f(vx, vy, wx, wy);
  if (l2 <= 1e-4f) { *cx = vx; *cy = vy; return; }
  float t = ((px - vx) * (wx - vx) + (py - vy) * (wy - vy)) / l2;
  if (t < 0.0f) { *cx = vx; *cy = vy; return; }
  if (t > 1.0f) { *cx = wx; *cy = wy; return; }
  *cx = vx + t * (wx - vx); *cy = vy + t * (wy - vy);
}

static int uf_find(int *lab, int a) { while (lab[a] != a) a = lab[a]; return a; }
static void uf_union(int *lab, int a, int b) {
  a = uf_find(lab, a); b = uf_find(lab, b);
  if (a < b) lab[b] = a; else if (b < a) lab[a] = b;
}

typedef struct { int64_t mx00, mx01, mx11, my0, my1; int16_t dx, dy, vx, vy; int32_t d2, pad; } lsx_t; /* pl:41-45 */

void rdo_polyline(void *lslist, int lslist_bytes, int *ids, const int *in, int ring_nonzero, float minerror, int sizeThre,
                  int iw, int ih, rdo_poly_dbg_t *dbg) {
  const int N = iw * ih;
  int *A = (int *)malloc(sizeof(int) * N), *B = (int *)malloc(sizeof(int) * N), *C = (int *)malloc(sizeof(int) * N);
  int *chain = (int *)malloc(sizeof(int) * N), *L = (int *)malloc(sizeof(int) * N);
  int *nx[2], *pv[2], *flag = (int *)malloc(sizeof(int) * N);
  for (int k = 0; k < 2; k++) { nx[k] = (int *)malloc(sizeof(int) * N); pv[k] = (int *)malloc(sizeof(int) * N); }
  int *num[2], *link[2];
  for (int k = 0; k < 2; k++) { num[k] = (int *)malloc(sizeof(int) * N); link[k] = (int *)malloc(sizeof(int) * N); }

  /* oclpolyline.c:222-235 tidy: junction counts, bridge 1-px gaps, two checkerboard thinning sweeps, cut branches */
  rdo_junction(A, in, 1, iw, ih);
  connect_poly(B, A, ring_nonzero ? 1 : 0, iw, ih);
  if (dbg && dbg->connect) memcpy(dbg->connect, B, sizeof(int) * N);
  rdo_stringify(A, B, 0, iw, ih);
  rdo_stringify(B, A, 1, iw, ih);
  remove_branch(chain, B, iw, ih);

  /* :237-248 chain components, open closed loops at their root pixel */
  rdo_label8(L, chain, 0, iw, ih);
  rdo_junction(A, chain, 1, iw, ih);
  memset(B, 0, sizeof(int) * N);
  for (int y = 1; y < ih - 1; y++)
    for (int x = 1; x < iw - 1; x++) { const int p = y * iw + x; if (A[p] == 2) B[L[p]]++; }           /* pl:149-155 */
  for (int y = 1; y < ih - 1; y++)
    for (int x = 1; x < iw - 1; x++) {
      const int p = y * iw + x;
      if (L[p] == p && B[p] == 0) { chain[p] = 0; L[p] = -1; }                                           /* pl:157-167 */
    }
  if (dbg && dbg->chain) memcpy(dbg->chain, chain, sizeof(int) * N);
  if (dbg && dbg->chain_label) memcpy(dbg->chain_label, L, sizeof(int) * N);

  /* :250-251 pl:193-220 immediate neighbours + orientation flags */
  for (int p = 0; p < N; p++) nx[0][p] = pv[0][p] = flag[p] = -1;
  for (int y = 1; y < ih - 1; y++)
    for (int x = 1; x < iw - 1; x++) {
      const int p = y * iw + x;
      if (L[p] == -1) continue;
      int a, b, f = 0;
      two_neighbours(L, p, iw, &a, &b);
      nx[0][p] = a; pv[0][p] = b;
      if (a != p) { int a2, b2; two_neighbours(L, a, iw, &a2, &b2); if (a2 == p) f |= 1; }
      if (b != p) { int a2, b2; two_neighbours(L, b, iw, &a2, &b2); if (b2 == p) f |= 2; }
      flag[p] = f;
    }

  /* :253-263 pl:222-267 four rounds of 8-hop pointer jumping towards both chain ends */
  for (int round = 0; round < 4; round++) {
    const int page = round & 1;
    const int *ni = nx[page], *pi = pv[page];
    int *no = nx[page ^ 1], *po = pv[page ^ 1];
    int *newflag = (int *)malloc(sizeof(int) * N);
    memcpy(newflag, flag, sizeof(int) * N);
    for (int p = 0; p < N; p++) no[p] = po[p] = -1;
    for (int y = 1; y < ih - 1; y++)
      for (int x = 1; x < iw - 1; x++) {
        const int p = y * iw + x;
        if (L[p] == -1) continue;
        int revn = page == 0 ? (flag[p] & 1) != 0 : (flag[p] & 4) != 0;
        int revp = page == 0 ? (flag[p] & 2) != 0 : (flag[p] & 8) != 0;
        int nn = ni[p], pp = pi[p];
        for (int i = 0; i < 8; i++) {
          int nn2 = revn ? pi[nn] : ni[nn];
          int pp2 = revp ? ni[pp] : pi[pp];
          int nf = flag[nn], pf = flag[pp];
          if (page != 0) { nf >>= 2; pf >>= 2; }
          revn = revn ? ((nf & 2) == 0) : ((nf & 1) != 0);
          revp = revp ? ((pf & 1) == 0) : ((pf & 2) != 0);
          nn = nn2; pp = pp2;
        }
        no[p] = nn; po[p] = pp;
        int f = flag[p];
        if (page == 0) { f &= 3; f |= revn ? 4 : 0; f |= revp ? 8 : 0; }
        else { f &= (3 << 2); f |= revn ? 1 : 0; f |= revp ? 2 : 0; }
        newflag[p] = f;
      }
    /* the kernel updates flags in place, but a launch only reads the bit pair it does not write */
    memcpy(flag, newflag, sizeof(int) * N);
    free(newflag);
  }
  /* after 4 rounds the ends are back in page 0 (tmp0/tmp2 at oclpolyline.c:265) */

  /* :265-266 pl:269-285 link every pixel towards the chain end with the smaller index */
  for (int p = 0; p < N; p++) { num[0][p] = 0; link[0][p] = -1; }
  for (int y = 1; y < ih - 1; y++)
    for (int x = 1; x < iw - 1; x++) {
      const int p = y * iw + x;
      if (L[p] == -1) continue;
      int a, b;
      two_neighbours(L, p, iw, &a, &b);
      link[0][p] = nx[0][p] < pv[0][p] ? a : b;
      num[0][p] = link[0][p] == p ? 0 : 1;
    }

  /* :268-275 pl:287-310 three rounds of 32-hop pointer-jumping prefix sums */
  for (int round = 0; round < 3; round++) {
    const int s = round & 1, d = s ^ 1;
    for (int p = 0; p < N; p++) { num[d][p] = 0; link[d][p] = -1; }
    for (int y = 1; y < ih - 1; y++)
      for (int x = 1; x < iw - 1; x++) {
        const int p = y * iw + x;
        if (link[s][p] == -1) { num[d][p] = num[s][p]; link[d][p] = -1; continue; }
        int no = num[s][p], lo = link[s][p], ok = 1;
        for (int i = 0; i < 32; i++) {
          if (!(0 < lo && lo < N)) { ok = 0; break; }
          no += num[s][lo];
          lo = link[s][lo];
        }
        if (ok) { num[d][p] = no; link[d][p] = lo; }
      }
  }
  const int *number = num[1];   /* three rounds: 0->1->0->1 */
  if (dbg && dbg->num) memcpy(dbg->num, number, sizeof(int) * N);

  /* :277-280 pl:312-355 split chains where the numbering is not continuous (run to convergence) */
  int *pix = A, *lab2 = B;
  for (int p = 0; p < N; p++) { lab2[p] = number[p] == 0 ? 0 : p; pix[p] = number[p] == 0 ? 0 : number[p] + 1; }
  for (int y = 1; y < ih - 1; y++)
    for (int x = 1; x < iw - 1; x++) {
      const int p = y * iw + x;
      if (pix[p] == 0) continue;
      for (int i = 0; i < 8; i++) {
        const int q = p + RX[i] + RY[i] * iw;
        if (pix[q] == 0) continue;
        const unsigned df = pix[p] > pix[q] ? (unsigned)pix[p] - (unsigned)pix[q] : (unsigned)pix[q] - (unsigned)pix[p];
        if (df <= 1) uf_union(lab2, p, q);
      }
    }
  for (int p = 0; p < N; p++) if (pix[p] != 0) lab2[p] = uf_find(lab2, p);
  if (dbg && dbg->sub_label) memcpy(dbg->sub_label, lab2, sizeof(int) * N);

  /* :282-288 pl:357-378 drop short chains */
  memset(C, 0, sizeof(int) * N);
  for (int p = 0; p < N; p++) if (lab2[p] != 0) C[lab2[p]]++;
  for (int p = 0; p < N; p++) ids[p] = C[lab2[p]] > sizeThre ? lab2[p] : 0;

  /* :290-295 pl:380-420 compact ids, handed out in raster order of the root pixels */
  memset(C, 0, sizeof(int) * N);
  int nchains = 0;
  for (int y = 1; y < ih - 1; y++)
    for (int x = 1; x < iw - 1; x++) { const int p = y * iw + x; if (ids[p] != 0 && ids[p] == p) C[p] = ++nchains; }
  for (int y = 0; y < ih; y++)
    for (int x = 0; x < iw; x++) {
      const int p = y * iw + x;
      ids[p] = interior(x, y, iw, ih) && ids[p] != 0 ? C[ids[p]] : 0;
    }
  if (dbg && dbg->ids0) memcpy(dbg->ids0, ids, sizeof(int) * N);

  /* ---------------------------------------------------------------- :297 mkpl, pl:439-646 */
  rdo_ls_t *ls = (rdo_ls_t *)lslist;
  int *hdr = (int *)lslist;
  const int cap = lslist_bytes;
  memset(lslist, 0, (size_t)lslist_bytes);
#define FITS(g) ((g) >= 0 && (long long)cap > (long long)((g) + 1) * (long long)sizeof(rdo_ls_t))

  for (int y = 1; y < ih - 1; y++)                                                                       /* pl:439-472 */
    for (int x = 1; x < iw - 1; x++) {
      const int p = y * iw + x, g = ids[p], n = number[p];
      if (g == 0 || !FITS(g)) continue;
      if (n == 1) { ls[g].x0 = (float)x; ls[g].y0 = (float)y; ls[g].level = 0; ls[g].startCount++; }
      ls[g].npix++;
      if (n > ls[g].endIndex) ls[g].endIndex = n;
      if (g > hdr[0]) hdr[0] = g;
    }
  for (int y = 1; y < ih - 1; y++)                                                                       /* pl:475-506 */
    for (int x = 1; x < iw - 1; x++) {
      const int p = y * iw + x, g = ids[p], n = number[p];
      if (g == 0 || !FITS(g)) continue;
      if (n != ls[g].endIndex) continue;
      if (ls[g].startCount == 1 && ls[g].npix >= 2) {
        if (ls[g].endCount++ == 0) { ls[g].x1 = (float)x; ls[g].y1 = (float)y; ls[g].polyid = g; }
      } else ls[g].polyid = 0;
    }

  int *dist = C;
  void *old = malloc((size_t)lslist_bytes);
  int moved_prev = 1;
  for (int it = 1; it <= 15 && moved_prev; it++) {
    /* pass 1: distance of every chain pixel to its segment's chord, per-segment maximum (pl:509-540) */
    for (int p = 0; p < N; p++) {
      const int g = ids[p];
      if (g == 0 || !FITS(g) || ls[g].polyid == 0) continue;
      const int x = p % iw, y = p / iw;
      const int x0 = (int)ls[g].x0, y0 = (int)ls[g].y0, x1 = (int)ls[g].x1, y1 = (int)ls[g].y1;
      float cx, cy;
      closest_on_segment((float)x0, (float)y0, (float)x1, (float)y1, (float)x, (float)y, &cx, &cy);
      const float a = cx - (float)x, b = cy - (float)y;
      int d = (int)((float)sqrt((double)a * (double)a + (double)b * (double)b) * 65536);
      d ^= pixel_rand(p, 0) & 0x1fff;
      dist[p] = d;
      if (d > ls[g].maxDist) ls[g].maxDist = d;
    }
    /* pass 2: split at the farthest pixel; reads the snapshot, writes the live list (pl:543-615) */
    memcpy(old, lslist, (size_t)lslist_bytes);
    const rdo_ls_t *gp = (const rdo_ls_t *)old;
    for (int p = 0; p < N; p++) {
      const int g = ids[p], n = number[p];
      if (g == 0 || !FITS(g) || gp[g].polyid == 0) continue;
      if (gp[g].endIndex - gp[g].startIndex < 3) continue;
      if (gp[g].startCount > 1 || gp[g].endCount > 1) continue;
      const int md = gp[g].maxDist;
      if (dist[p] != md) continue;
      if (md < (int)(minerror * 65536)) continue;
      if ((float)md < (minerror * 3 * 65536) && (float)md * (float)md / dist2f(gp[g].x0, gp[g].y0, gp[g].x1, gp[g].y1) < 100000.0f) continue;
      const int x = p % iw, y = p / iw;
      if (dist2f((float)x, (float)y, gp[g].x0, gp[g].y0) < 1) continue;
      if (dist2f((float)x, (float)y, gp[g].x1, gp[g].y1) < 1) continue;
      const int gr = gp[g].rightPtr;
      const int gn = ++hdr[0];
      if (!FITS(gn)) continue;
      ls[gn].startIndex = n; ls[gn].endIndex = gp[g].endIndex;
      ls[gn].x0 = (float)x; ls[gn].y0 = (float)y; ls[gn].x1 = gp[g].x1; ls[gn].y1 = gp[g].y1;
      ls[gn].leftPtr = g; ls[gn].rightPtr = gp[g].rightPtr;
      ls[gn].maxDist = 0; ls[gn].polyid = gp[g].polyid; ls[gn].level = md;
      ls[g].endIndex = n; ls[g].x1 = (float)x; ls[g].y1 = (float)y; ls[g].rightPtr = gn; ls[g].maxDist = 0;
      if (gr != 0) ls[gr].leftPtr = gn;
    }
    /* pass 3: pixels beyond the new end move to the right-hand segment (pl:618-646) */
    int moved = 0;
    for (int p = 0; p < N; p++) {
      const int g = ids[p];
      if (g == 0 || !FITS(g) || ls[g].polyid == 0) continue;
      if (ls[g].endIndex < number[p]) { ids[p] = ls[g].rightPtr; moved = 1; }
    }
    moved_prev = moved;
  }
  free(old);

  /* ---------------------------------------------------------------- :299-306 refine, pl:680-809 */
  const int nseg = hdr[0];
  lsx_t *sx = (lsx_t *)calloc((size_t)nseg + 2, sizeof(lsx_t));
  for (int g = 1; g <= nseg; g++) {
    if (ls[g].polyid == 0) continue;
    sx[g].dx = (int16_t)(ls[g].x1 - ls[g].x0); sx[g].dy = (int16_t)(ls[g].y1 - ls[g].y0);
    sx[g].vx = (int16_t)-sx[g].dy; sx[g].vy = sx[g].dx;
    sx[g].d2 = sx[g].dx * sx[g].dx + sx[g].dy * sx[g].dy;
  }
  for (int p = 0; p < N; p++) {
    const int g = ids[p];
    if (g == 0 || g < 0 || nseg < g) continue;
    if (ls[g].polyid == 0) continue;   /* the kernel also accumulates for invalid records; those sums are never read */
    const int x = p % iw, y = p / iw;
    const int vx = x - (int)rintf(ls[g].x0), vy = y - (int)rintf(ls[g].y0);
    const int ay = vx * sx[g].vx + vy * sx[g].vy;
    const int ax0 = vx * sx[g].dx + vy * sx[g].dy;
    const int ax1 = sx[g].d2;
    sx[g].mx00 += (int64_t)rintf((float)ax0 * (float)ax0);
    sx[g].mx01 += (int64_t)rintf((float)ax0 * (float)ax1);
    sx[g].mx11 += (int64_t)rintf((float)ax1 * (float)ax1);
    sx[g].my0 += (int64_t)rintf((float)ax0 * (float)ay);
    sx[g].my1 += (int64_t)rintf((float)ax1 * (float)ay);
  }
  for (int g = 1; g <= nseg; g++) {                                                                       /* pl:752-770 */
    if (ls[g].polyid == 0) continue;
    float rdet = (float)sx[g].mx00 * (float)sx[g].mx11 - (float)sx[g].mx01 * (float)sx[g].mx01;
    if (rdet == 0) continue;
    rdet = (float)(1.0 / (double)rdet);
    const float as0 = ((float)sx[g].mx11 * (float)sx[g].my0 - (float)sx[g].mx01 * (float)sx[g].my1) * rdet;
    const float as1 = ((float)sx[g].mx00 * (float)sx[g].my1 - (float)sx[g].mx01 * (float)sx[g].my0) * rdet;
    ls[g].x0 += (float)sx[g].vx * as1; ls[g].y0 += (float)sx[g].vy * as1;
    ls[g].x1 += (float)sx[g].vx * (as0 + as1); ls[g].y1 += (float)sx[g].vy * (as0 + as1);
  }
  for (int g = 1; g <= nseg; g++) {                                                                       /* pl:772-809, ascending g (H15) */
    if (ls[g].polyid == 0) continue;
    const int h = ls[g].rightPtr;
    if (h == 0) continue;
    const float v0 = ls[g].x0, v1 = ls[g].y0, v2 = ls[g].x1, v3 = ls[g].y1;
    const float u0 = ls[h].x0, u1 = ls[h].y0, u2 = ls[h].x1, u3 = ls[h].y1;
    const float d = (v2 - v0) * (u3 - u1) - (v3 - v1) * (u2 - u0);
    float wx, wy;
    if ((double)fabsf(d) < 1e-6) {
      wx = (v2 + u0) * 0.5f; wy = (v3 + u1) * 0.5f;
    } else {
      const float n = (v1 - u1) * (u2 - u0) - (v0 - u0) * (u3 - u1);
      const float q = n / d;
      wx = v0 + q * (v2 - v0); wy = v1 + q * (v3 - v1);
      const float e0 = sqrtf((wx - v2) * (wx - v2) + (wy - v3) * (wy - v3));
      const float e1 = sqrtf((wx - u0) * (wx - u0) + (wy - u1) * (wy - u1));
      if (e0 > 10 && e1 > 10) { wx = (v2 + u0) * 0.5f; wy = (v3 + u1) * 0.5f; }
    }
    ls[g].x1 = wx; ls[g].y1 = wy; ls[h].x0 = wx; ls[h].y0 = wy;
  }
  free(sx);
#undef FITS

  free(A); free(B); free(C); free(chain); free(L); free(flag);
  for (int k = 0; k < 2; k++) { free(nx[k]); free(pv[k]); free(num[k]); free(link[k]); }
}

/* poly.cpp:104-123 */
void rdo_poly_frame(void *lslist, int *ids, const uint8_t *bgr, int iw, int ih, int ws, int strengthThre, float minerror, int sizeThre) {
  const int N = iw * ih;
  uint32_t *plab = (uint32_t *)malloc(4 * (size_t)N);
  float *f0 = (float *)malloc(4 * (size_t)N), *f1 = (float *)malloc(4 * (size_t)N), *f2 = (float *)malloc(4 * (size_t)N);
  float *g0 = (float *)malloc(4 * (size_t)N), *g1 = (float *)malloc(4 * (size_t)N), *g2 = (float *)malloc(4 * (size_t)N);
  float *vxy = (float *)malloc(8 * (size_t)N);
  int *m = (int *)malloc(4 * (size_t)N), *lab = (int *)malloc(4 * (size_t)N), *str = (int *)calloc(N, 4);
  rdo_bgr2plab(plab, bgr, iw, ih, ws);
  rdo_unpack_plab(f0, f1, f2, plab, N);
  rdo_iirblur(g0, f0, iw, ih); rdo_iirblur(g1, f1, iw, ih); rdo_iirblur(g2, f2, iw, ih);
  rdo_pack_plab(plab, g0, g1, g2, N);
  rdo_edgevec(vxy, g0, iw, ih);
  rdo_edge_plab(f0, plab, iw, ih);
  rdo_thinthres(f1, f0, vxy, iw, ih);
  rdo_positive_mask(m, f1, N);
  rdo_label8(lab, m, 0, iw, ih);
  rdo_calc_strength(str, f1, lab, iw, ih);
  rdo_filter_strength(lab, str, strengthThre, iw, ih);
  rdo_threshold_i(m, lab, 0, 0, 1, N);
  rdo_polyline(lslist, N * 16, ids, m, 0, minerror, sizeThre, iw, ih, NULL);
  free(plab); free(f0); free(f1); free(f2); free(g0); free(g1); free(g2); free(vxy); free(m); free(lab); free(str);
}
