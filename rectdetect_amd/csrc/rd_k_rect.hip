// rectdetect-mi355x: rect-path kernels for gfx950 - edge tidy, edge-stopped blur, quantise/despeckle, merge mask,
// region labelling, boundary marking, segment/boundary voting and result sampling.
//
// Reference behaviour being reproduced: oclrect.cl ("rc"), oclrect.c ("rh").  See rd_device.h for arithmetic rules.
#include "rd_device.h"
#include "rd_kernels.h"

namespace {

using namespace rd;

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
const dim3 block2(64, 4);
inline dim3 grid2(int iw, int ih) { return dim3(cdiv(iw, 64), cdiv(ih, 4)); }
inline int ew_grid(int n) { int g = cdiv(n, 256 * 4); return g < 1 ? 1 : (g > 4096 ? 4096 : g); }

#define RD_XY const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y

// ------------------------------------------------------------------------------------------------ edge tidy
// rc:74-95 (`> 0`) and pl:66-87 (`!= 0`): 3x3 population count of on-pixels, 1 -> 0, frame border 0
__global__ __launch_bounds__(256) void k_junction(int *__restrict__ out, const int *__restrict__ in, int nz, int iw, int ih) {
  RD_XY;
  if (x >= iw || y >= ih) return;
  const int p = y * iw + x;
  int r = 0;
  if (x > 0 && y > 0 && x < iw - 1 && y < ih - 1) {
    const int c = in[p];
    if (nz ? c != 0 : c > 0) {
      int count = 1;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int q = in[p + nbr_dx(i) + nbr_dy(i) * iw];
        if (nz ? q != 0 : q > 0) count++;
      }
      r = count == 1 ? 0 : count;
    }
  }
  out[p] = r;
}

// rc:97-121: on-pixels stay, 1-px gaps next to a curve end (count == 2) are closed by ten patterns; 2-px ring -> 0
__global__ __launch_bounds__(256) void k_connect_rect(int *__restrict__ out, const int *__restrict__ in, int iw, int ih) {
  RD_XY;
  if (x >= iw || y >= ih) return;
  const int p = y * iw + x;
  int o = 0;
  if (x > 1 && y > 1 && x < iw - 2 && y < ih - 2) {
    if (in[p] != 0) o = 1;
    else {
      const int w = in[p - 1], e = in[p + 1], n = in[p - iw], s = in[p + iw];
      const int nw = in[p - iw - 1], ne = in[p - iw + 1], sw = in[p + iw - 1], se = in[p + iw + 1];
      if (w == 2 && e != 0) o = 1;
      if (w != 0 && e == 2) o = 1;
      if (n == 2 && s != 0) o = 1;
      if (n != 0 && s == 2) o = 1;
      if (nw == 2 && se == 2) o = 1;
      if (ne == 2 && sw == 2) o = 1;
      if (e == 2 && sw == 2) o = 1;
      if (w == 2 && se == 2) o = 1;
      if (ne == 2 && s == 2) o = 1;
      if (nw == 2 && s == 2) o = 1;
    }
  }
  out[p] = o;
}

// rc:123-135 == pl:112-124: checkerboard thinning - a pixel of parity mod2 with an orthogonal L-shaped pair of
// on-neighbours is removed; everything else (including the frame border) is copied
__global__ __launch_bounds__(256) void k_stringify(int *__restrict__ out, const int *__restrict__ in, int mod2, int iw, int ih) {
  RD_XY;
  if (x >= iw || y >= ih) return;
  const int p = y * iw + x;
  int v = in[p];
  if (x > 0 && y > 0 && x < iw - 1 && y < ih - 1 && ((x + y) & 1) == mod2) {
    const bool up = in[p - iw] != 0, dn = in[p + iw] != 0, lf = in[p - 1] != 0, rt = in[p + 1] != 0;
    if ((up || dn) && (lf || rt)) v = 0;
  }
  out[p] = v;
}

// ------------------------------------------------------------------------------------------------ edge-stopped box blur
// rc:155-205: mean of the packed-Lab integer fields over up to 4 pixels on either side along one axis (centre counted
// twice), each side stopping at transitions of the int8 edge mask.  VERT selects the axis.
template <int VERT>
__global__ __launch_bounds__(256) void k_blblur(uint32_t *__restrict__ out, const int8_t *__restrict__ edge, const uint32_t *__restrict__ in, int iw, int ih) {
  RD_XY;
  if (x >= iw || y >= ih) return;
  const int p = y * iw + x;
  const int c0 = VERT ? y : x, n = VERT ? ih : iw, st = VERT ? iw : 1;
  const int base = VERT ? x : y * iw;
  const bool has_side = VERT ? (x < iw - 1) : (y < ih - 1);
  const int side = VERT ? 1 : iw;
  int wsum = 0, s0 = 0, s1 = 0, s2 = 0;
  const bool oe = edge[p] != 0;
  for (int d = 0; d >= -4; d--) {
    const int c = c0 + d;
    if (c < 0) break;
    const int q = base + c * st;
    const int ec = edge[q];
    if (c > 0) {
      const int em = edge[q - st];
      if (ec != 0 && em == 0) break;
      if (has_side && ec == 0 && em != 0 && edge[q + side] != 0) break;
    }
    wsum++;
    const uint32_t v = in[q];
    s0 += v & 4095; s1 += (v >> 12) & 1023; s2 += (v >> 22) & 1023;
  }
  for (int d = 0; d <= 4; d++) {
    const int c = c0 + d;
    if (c > n - 1) break;
    const int q = base + c * st;
    const int ec = edge[q];
    if (c < n - 1 && ec == 0 && edge[q + st] != 0) break;
    if (oe && ec == 0) break;
    wsum++;
    const uint32_t v = in[q];
    s0 += v & 4095; s1 += (v >> 12) & 1023; s2 += (v >> 22) & 1023;
  }
  uint32_t r = in[p];
  if (wsum != 0) {
    r = (uint32_t)clampi(s2 / wsum, 0, 1023);
    r = (r << 10) | (uint32_t)clampi(s1 / wsum, 0, 1023);
    r = (r << 12) | (uint32_t)clampi(s0 / wsum, 0, 4095);
  }
  out[p] = r;
}

// rc:207-216
__global__ void k_quantize(uint32_t *__restrict__ out, const uint32_t *__restrict__ in, int n0, int n1, int n2, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float L, a, b;
    unpack_lab(in[i], L, a, b);
    out[i] = pack_lab(roundf(L * n0) / (float)n0, roundf(a * n1) / (float)n1, roundf(b * n2) / (float)n2);
  }
}

// rc:218-244: pixels with a non-zero NMS response take the colour of the Lab-nearest 3x3 neighbour without one
__global__ __launch_bounds__(256) void k_despeckle(uint32_t *__restrict__ out, const uint32_t *__restrict__ in, const float *__restrict__ edge, int iw, int ih) {
  RD_XY;
  if (x >= iw || y >= ih) return;
  const int p0 = y * iw + x;
  uint32_t r = in[p0];
  if (!(edge[p0] < 1e-6f)) {
    float dist = 1e+10f, l0, a0, b0;
    unpack_lab(r, l0, a0, b0);
    for (int yy = -1; yy <= 1; yy++)
      for (int xx = -1; xx <= 1; xx++) {
        if (x + xx < 0 || x + xx >= iw || y + yy < 0 || y + yy >= ih) continue;
        const int p1 = p0 + yy * iw + xx;
        if (edge[p1] >= 1e-6f) continue;
        float l1, a1, b1;
        const uint32_t v = in[p1];
        unpack_lab(v, l1, a1, b1);
        const float dx = l1 - l0, dy = a1 - a0, dz = b1 - b0;
        const float d = sqrtf(dx * dx + dy * dy + dz * dz);
        if (d < dist) { r = v; dist = d; }
      }
  }
  out[p0] = r;
}

// ------------------------------------------------------------------------------------------------ merge mask
// rc:246-287.  pass 0: every pixel with a non-zero junction count sets the ring 16 <= d^2 < 36 around itself;
// pass 1: curve ends (count 2) erase the disc d^2 < 64, all other counted pixels the disc d^2 < 16.  The writes of one
// pass all store the same value, so their order is irrelevant; the two passes are separate launches.
__global__ __launch_bounds__(256) void k_merge_mask(int *out, const int *__restrict__ junction, int pass, int iw, int ih) {
  RD_XY;
  if (x >= iw || y >= ih) return;
  const int j = junction[y * iw + x];
  if (j == 0) return;
  const int r = pass == 0 ? 6 : (j == 2 ? 8 : 4);
  const int lim = j == 2 ? 64 : 16;
  for (int yy = y - r; yy <= y + r; yy++) {
    if (yy < 0 || yy >= ih) continue;
    for (int xx = x - r; xx <= x + r; xx++) {
      if (xx < 0 || xx >= iw) continue;
      const int d2 = (yy - y) * (yy - y) + (xx - x) * (xx - x);
      if (pass == 0) { if (16 <= d2 && d2 < 36) out[yy * iw + xx] = 1; }
      else if (d2 < lim) out[yy * iw + xx] = 0;
    }
  }
}

// ------------------------------------------------------------------------------------------------ regions
// rc:289-298 initial links (up if same colour, else left if same colour, else self)
__global__ __launch_bounds__(256) void k_region_init(int *__restrict__ label, const int *__restrict__ pix, int iw, int ih) {
  RD_XY;
  if (x >= iw || y >= ih) return;
  const int p = y * iw + x;
  const int v = pix[p];
  int l = p;
  if (y > 0 && v == pix[p - iw]) l = p - iw;
  else if (x > 0 && v == pix[p - 1]) l = p - 1;
  label[p] = l;
}

// rc:300-334 per-pixel rule, evaluated in SYNCHRONOUS rounds: every pixel reads the labels of the previous round,
// proposes `min` updates for itself and for its old parent, and the proposals are applied between rounds.  The
// reference applies the same rule in place for 8 launches, which makes its result depend on the work-item order
// (SURVEY.md H5); synchronous rounds to convergence are the order-free reading of the same rule.
__global__ __launch_bounds__(256) void k_region_propose(const int *__restrict__ label, int *prop, const int *__restrict__ pix, const int *__restrict__ mask,
                                                        const int *__restrict__ edge, int iw, int ih, const int *flags, int round) {
  if (round > 0 && flags[round - 1] == 0) return;
  RD_XY;
  const bool inside = x > 0 && y > 0 && x < iw - 1 && y < ih - 1;
  int og = 0, g = 0, p0 = 0;
  if (inside) {
    p0 = y * iw + x;
    og = label[p0];
    g = og;
    const int c = pix[p0];
    const bool any = mask[p0] != 0;
    const bool e0 = edge[p0] <= 0;
    int p1, s;
    p1 = p0 - iw; s = label[p1]; if (s < g && (c == pix[p1] || any) && e0) g = s;
    p1 = p0 - 1;  s = label[p1]; if (s < g && (c == pix[p1] || any) && e0) g = s;
    p1 = p0 + 1;  s = label[p1]; if (s < g && (c == pix[p1] || any) && edge[p1] <= 0) g = s;
    p1 = p0 + iw; s = label[p1]; if (s < g && (c == pix[p1] || any) && edge[p1] <= 0) g = s;
    for (int j = 0; j < 8; j++) g = label[g];   // rc:328: eight pointer jumps
  }
  bool todo = inside && g != og;
  if (todo && g < prop[p0]) atomicMin(&prop[p0], g);
  // the old parent is shared by (up to) a whole region: one atomic per distinct parent per wave, and only if it can
  // still lower the stored proposal (a stale read only costs a redundant atomic)
  while (__any(todo)) {
    const unsigned long long m = __ballot(todo);
    const int leader = __ffsll((long long)m) - 1;
    const int lo = __shfl(og, leader);
    const bool mine = todo && og == lo;
    int v = mine ? g : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    if ((int)threadIdx.x == leader && v < prop[lo]) atomicMin(&prop[lo], v);
    if (mine) todo = false;
  }
}

__global__ void k_region_apply(int *label, int *prop, int n, int *flags, int round) {
  if (round > 0 && flags[round - 1] == 0) return;
  bool changed = false;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int m = prop[i];
    if (m != 0x7f7f7f7f) {
      prop[i] = 0x7f7f7f7f;
      if (m < label[i]) { label[i] = m; changed = true; }
    }
  }
  if (__any(changed) && (threadIdx.x & 63) == 0) flags[round] = 1;
}

// rc:336-346: out[label]++ for every pixel.  Most pixels belong to a handful of huge regions, so counts are
// aggregated per wave (ballot of equal labels), then per block in an LDS hash, before touching global atomics.
#define RS_T 1024
__global__ __launch_bounds__(256) void k_region_size(int *out, const int *__restrict__ label, int n) {
  __shared__ int keys[RS_T], vals[RS_T];
  for (int i = threadIdx.x; i < RS_T; i += 256) { keys[i] = -1; vals[i] = 0; }
  __syncthreads();
  const int per_block = 256 * 32;
  const int begin = blockIdx.x * per_block;
  for (int k = 0; k < 32; k++) {
    const int i = begin + k * 256 + threadIdx.x;
    int l = i < n ? label[i] : -1;
    bool todo = l != -1;
    while (__any(todo)) {
      const unsigned long long m = __ballot(todo);
      const int leader = __ffsll((long long)m) - 1;
      const int ll = __shfl(l, leader);
      const bool mine = todo && l == ll;
      const int cnt = __popcll(__ballot(mine));
      if ((int)(threadIdx.x & 63) == leader) {
        unsigned h = ((unsigned)ll * 2654435761u) >> 22;
        int probes = 0;
        for (;;) {
          const int kprev = atomicCAS(&keys[h], -1, ll);
          if (kprev == -1 || kprev == ll) { atomicAdd(&vals[h], cnt); break; }
          h = (h + 1) & (RS_T - 1);
          if (++probes == 32) { atomicAdd(&out[ll], cnt); break; }
        }
      }
      if (mine) todo = false;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < RS_T; i += 256)
    if (keys[i] != -1 && vals[i] != 0) atomicAdd(&out[keys[i]], vals[i]);
}

// rc:348-371.  The reference updates labels in place in raster order: a small-region pixel sees the NEW labels of
// its NW, N, NE, W neighbours.  One Jacobi round of that recurrence: `cur` holds the previous round's
// values for small-region pixels; the rounds are iterated by the launcher.
__global__ __launch_bounds__(256) void k_despeckle2_round(int *__restrict__ nxt, const int *__restrict__ cur, const int *__restrict__ old, const int *__restrict__ size,
                                                           int thre, int iw, int ih) {
  RD_XY;
  if (x >= iw || y >= ih) return;
  const int p0 = y * iw + x;
  const int l0 = old[p0];
  int res = l0;
  if (size[l0] <= thre) {
    int maxSize = 0;
    for (int yy = -1; yy <= 1; yy++)
      for (int xx = -1; xx <= 1; xx++) {
        if (x + xx < 0 || x + xx >= iw || y + yy < 0 || y + yy >= ih) continue;
        const int p1 = p0 + yy * iw + xx;
        const int l1 = (yy < 0 || (yy == 0 && xx < 0)) ? cur[p1] : old[p1];
        const int sz = size[l1];
        if (sz > maxSize) { maxSize = sz; res = l1; }
      }
  }
  nxt[p0] = res;
}

// rc:373-390
__global__ __launch_bounds__(256) void k_mark_boundary(int *__restrict__ out, const int *__restrict__ in, int iw, int ih) {
  RD_XY;
  if (x >= iw || y >= ih) return;
  const int p0 = y * iw + x;
  int r = -1;
  if (x > 1 && y > 1 && x < iw - 2 && y < ih - 2) {
    const int c0 = in[p0];
    bool near = false;
#pragma unroll
    for (int yy = -2; yy <= 2; yy++)
#pragma unroll
      for (int xx = -2; xx <= 2; xx++) near = near || (in[p0 + yy * iw + xx] != c0);
    if (near) r = c0;
  }
  out[p0] = r;
}

// ------------------------------------------------------------------------------------------------ voting
// rc:426-464 in its canonical (raster) reading, SURVEY.md H9: a table slot is owned by the first pixel in raster order
// that touches it; every pixel whose segment id equals the owner's widens the slot's box by its own position, except
// that the owning pixel's very first touch does not.  Two launches: claim (atomicMin of pixel index per slot), then
// box (atomicMax), both over the pixels that carry a segment id.
__device__ __forceinline__ unsigned ls_slot(int id, int bid, int nentry) { return (((unsigned)id * (unsigned)bid) & 0x7fffffffu) % (unsigned)nentry; }

__global__ __launch_bounds__(256) void k_reduce_claim(int *claim, const int *__restrict__ boundary, const int *__restrict__ lsid, int iw, int ih, int nentry) {
  RD_XY;
  if (x <= 0 || y <= 0 || x >= iw - 1 || y >= ih - 1) return;
  const int p0 = y * iw + x, id = lsid[p0];
  if (id <= 0) return;
  for (int yy = -3; yy <= 3; yy++) {
    if (y + yy < 0 || y + yy >= ih) continue;
    for (int xx = -3; xx <= 3; xx++) {
      if (x + xx < 0 || x + xx >= iw) continue;
      const int b = boundary[(y + yy) * iw + x + xx];
      if (b <= 0) continue;
      atomicMin(&claim[ls_slot(id, b, nentry)], p0);
    }
  }
}

__global__ __launch_bounds__(256) void k_reduce_box(int *table, const int *__restrict__ claim, const int *__restrict__ boundary, const int *__restrict__ lsid, int iw, int ih, int nentry) {
  RD_XY;
  if (x <= 0 || y <= 0 || x >= iw - 1 || y >= ih - 1) return;
  const int p0 = y * iw + x, id = lsid[p0];
  if (id <= 0) return;
  // slots this pixel touches, in scan order; a slot can be touched several times (same or different boundary ids)
  unsigned first_claimed = 0xffffffffu;   // slot whose claiming touch (by this pixel) has been consumed
  bool have_first = false;
  for (int yy = -3; yy <= 3; yy++) {
    if (y + yy < 0 || y + yy >= ih) continue;
    for (int xx = -3; xx <= 3; xx++) {
      if (x + xx < 0 || x + xx >= iw) continue;
      const int b = boundary[(y + yy) * iw + x + xx];
      if (b <= 0) continue;
      const unsigned slot = ls_slot(id, b, nentry);
      const int owner_pixel = claim[slot];
      const int owner_id = lsid[owner_pixel];
      if (owner_id != id) continue;
      if (owner_pixel == p0) {
        // this pixel claimed the slot: its first touch of the slot only claims (a pixel can own several slots)
        bool seen = false;
        if (have_first && first_claimed == slot) seen = true;
        if (!seen) {
          // was an earlier touch of this same slot already consumed?  keep a tiny set: re-scan the window prefix
          bool earlier = false;
          for (int y2 = -3; y2 <= yy && !earlier; y2++) {
            if (y + y2 < 0 || y + y2 >= ih) continue;
            for (int x2 = -3; x2 <= 3; x2++) {
              if (y2 == yy && x2 >= xx) break;
              if (x + x2 < 0 || x + x2 >= iw) continue;
              const int b2 = boundary[(y + y2) * iw + x + x2];
              if (b2 > 0 && ls_slot(id, b2, nentry) == slot) { earlier = true; break; }
            }
          }
          if (!earlier) { first_claimed = slot; have_first = true; continue; }
        }
      }
      int *e = table + (size_t)slot * 5;
      atomicMax(&e[1], iw - x);
      atomicMax(&e[2], x);
      atomicMax(&e[3], ih - y);
      atomicMax(&e[4], y);
    }
  }
}

__global__ void k_reduce_owner(int *table, const int *__restrict__ claim, const int *__restrict__ lsid, int nentry) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nentry; i += gridDim.x * blockDim.x) {
    const int c = claim[i];
    if (c != 0x7f7f7f7f) table[(size_t)i * 5] = lsid[c];
  }
}

// ------------------------------------------------------------------------------------------------ result sampling
// rh:1066-1098: 3 points along each valid segment x 5 offsets along its normal; the boundary id under each probe and
// the voting-table slot of (segment, boundary id) are gathered on the device so that only n*15*6 ints travel to the
// host instead of the 4N-byte plane and the 16N-byte table.  Double precision, same operation order as the host code.
struct ls_rec { float x0, y0, x1, y1; int startIndex, endIndex, leftPtr, rightPtr, startCount, endCount, maxDist, polyid, npix, level; };

__global__ void k_sample_segments(int *__restrict__ out, const ls_rec *__restrict__ ls, int max_records, const int *__restrict__ boundary, const int *__restrict__ table,
                                  int iw, int ih, int nentry) {
  const int n = *(const int *)ls;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t / 15 + 1, k = t % 15;
  if (i > n || i >= max_records) return;
  int *o = out + (size_t)(i * 15 + k) * 6;
  int segid = 0;
  if (ls[i].polyid != 0) {
    const double x0 = rint((double)ls[i].x0), y0 = rint((double)ls[i].y0), x1 = rint((double)ls[i].x1), y1 = rint((double)ls[i].y1);
    const int j = k / 5, dist = k % 5 - 2;
    const double ex = x1 - x0, ey = y1 - y0;
    const double inv = 1.0 / (sqrt(ex * ex + ey * ey) + 1e-20);
    const double dx = ex * inv, dy = ey * inv;
    const double vdx = -dy, vdy = dx;
    const double f = (j + 0.5) / 3;
    const double px = x0 + ex * f, py = y0 + ey * f;
    const double cx = px + vdx * dist, cy = py + vdy * dist;
    const int sx = (int)(cx + 0.5), sy = (int)(cy + 0.5);
    if (!(sx < 0 || sx >= iw || sy < 0 || sy >= ih)) segid = boundary[sx + sy * iw];
  }
  o[0] = segid;
  if (segid > 0) {
    const unsigned slot = ls_slot(i, segid, nentry);
    const int *e = table + (size_t)slot * 5;
    o[1] = e[0]; o[2] = e[1]; o[3] = e[2]; o[4] = e[3]; o[5] = e[4];
  } else {
    o[1] = o[2] = o[3] = o[4] = o[5] = 0;
  }
}

}  // namespace

namespace rdk {

void junction(hipStream_t s, int *out, const int *in, int nonzero_variant, int iw, int ih) {
  hipLaunchKernelGGL(k_junction, grid2(iw, ih), block2, 0, s, out, in, nonzero_variant, iw, ih);
}
void connect_rect(hipStream_t s, int *out, const int *in, int iw, int ih) {
  hipLaunchKernelGGL(k_connect_rect, grid2(iw, ih), block2, 0, s, out, in, iw, ih);
}
void stringify(hipStream_t s, int *out, const int *in, int mod2, int iw, int ih) {
  hipLaunchKernelGGL(k_stringify, grid2(iw, ih), block2, 0, s, out, in, mod2, iw, ih);
}
void blblur(hipStream_t s, uint32_t *out, const int8_t *edge, const uint32_t *in, int vertical, int iw, int ih) {
  if (vertical) hipLaunchKernelGGL(k_blblur<1>, grid2(iw, ih), block2, 0, s, out, edge, in, iw, ih);
  else hipLaunchKernelGGL(k_blblur<0>, grid2(iw, ih), block2, 0, s, out, edge, in, iw, ih);
}
void quantize(hipStream_t s, uint32_t *out, const uint32_t *in, int n0, int n1, int n2, int n) {
  hipLaunchKernelGGL(k_quantize, dim3(ew_grid(n)), dim3(256), 0, s, out, in, n0, n1, n2, n);
}
void despeckle(hipStream_t s, uint32_t *out, const uint32_t *in, const float *edge, int iw, int ih) {
  hipLaunchKernelGGL(k_despeckle, grid2(iw, ih), block2, 0, s, out, in, edge, iw, ih);
}
void merge_mask(hipStream_t s, int *out, const int *junction, int iw, int ih) {
  (void)hipMemsetAsync(out, 0, sizeof(int) * (size_t)iw * ih, s);
  hipLaunchKernelGGL(k_merge_mask, grid2(iw, ih), block2, 0, s, out, junction, 0, iw, ih);
  hipLaunchKernelGGL(k_merge_mask, grid2(iw, ih), block2, 0, s, out, junction, 1, iw, ih);
}

// scratch: 2*N ints (proposal plane + round flags at the start of the second plane)
void region_merge(hipStream_t s, int *label, int *scratch, const int *pix, const int *mask, const int *edge, int iw, int ih) {
  const int n = iw * ih, ROUNDS = 32;
  int *prop = scratch, *flags = scratch + n;
  (void)hipMemsetAsync(prop, 0x7f, sizeof(int) * (size_t)n, s);
  (void)hipMemsetAsync(flags, 0, sizeof(int) * (ROUNDS + 1), s);
  hipLaunchKernelGGL(k_region_init, grid2(iw, ih), block2, 0, s, label, pix, iw, ih);
  for (int r = 0; r < ROUNDS; r++) {
    hipLaunchKernelGGL(k_region_propose, grid2(iw, ih), block2, 0, s, (const int *)label, prop, pix, mask, edge, iw, ih, (const int *)flags, r);
    hipLaunchKernelGGL(k_region_apply, dim3(ew_grid(n)), dim3(256), 0, s, label, prop, n, flags, r);
  }
}

void region_size(hipStream_t s, int *out, const int *label, int n) {
  hipLaunchKernelGGL(k_region_size, dim3(cdiv(n, 256 * 32)), dim3(256), 0, s, out, label, n);
}

// scratch: 2*N ints.  On return `label` holds the result.
void despeckle2(hipStream_t s, int *label, int *scratch, const int *size, int thre, int iw, int ih) {
  const int n = iw * ih, ROUNDS = 8;   // even: the last round writes back into `label`
  int *old = scratch, *tmp = scratch + n;
  (void)hipMemcpyAsync(old, label, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, s);
  const int *cur = old;
  for (int r = 0; r < ROUNDS; r++) {
    int *nxt = (r & 1) ? label : tmp;
    hipLaunchKernelGGL(k_despeckle2_round, grid2(iw, ih), block2, 0, s, nxt, cur, (const int *)old, size, thre, iw, ih);
    cur = nxt;
  }
}

void mark_boundary(hipStream_t s, int *out, const int *in, int iw, int ih) {
  hipLaunchKernelGGL(k_mark_boundary, grid2(iw, ih), block2, 0, s, out, in, iw, ih);
}

// table: nentry*5 ints (cleared here); claim: nentry ints of scratch
void reduce_ls(hipStream_t s, int *table, int *claim, const int *boundary, const int *lsid, int iw, int ih, int nentry) {
  (void)hipMemsetAsync(table, 0, sizeof(int) * 5 * (size_t)nentry, s);
  (void)hipMemsetAsync(claim, 0x7f, sizeof(int) * (size_t)nentry, s);
  hipLaunchKernelGGL(k_reduce_claim, grid2(iw, ih), block2, 0, s, claim, boundary, lsid, iw, ih, nentry);
  hipLaunchKernelGGL(k_reduce_owner, dim3(ew_grid(nentry)), dim3(256), 0, s, table, (const int *)claim, lsid, nentry);
  hipLaunchKernelGGL(k_reduce_box, grid2(iw, ih), block2, 0, s, table, (const int *)claim, boundary, lsid, iw, ih, nentry);
}

void sample_segments(hipStream_t s, int *out, const void *lslist, int max_records, const int *boundary, const int *table, int iw, int ih, int nentry) {
  const int threads = max_records * 15;
  hipLaunchKernelGGL(k_sample_segments, dim3(cdiv(threads, 256)), dim3(256), 0, s, out, (const ls_rec *)lslist, max_records, boundary, table, iw, ih, nentry);
}

}  // namespace rdk
