// rectdetect-mi355x: connected-component labelling and per-label strength reduction for gfx950.
//
// The reference labels components with a fixed number of in-place min-propagation passes
// (oclimgutil.cl:495-538, oclimgutil.c:227-246: 1 + 10 launches, converged only by luck - SURVEY.md H4).
// Here the labelling is a run-based union-find that always converges: label = smallest pixel index of the
// 8-connected component of equal pixel value, -1 for pixels equal to the background value.
#include "rd_device.h"
#include "rd_kernels.h"

namespace {

using namespace rd;

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
const dim3 block2(64, 4);
inline dim3 grid2(int iw, int ih) { return dim3(cdiv(iw, 64), cdiv(ih, 4)); }

// Phase 1: every pixel points at the first pixel of its horizontal run inside the wave's 64-pixel row segment
// (ballot of run starts + count-leading-zeros); runs are the unit the merge phase works on.
__global__ __launch_bounds__(256) void k_label_init(int *__restrict__ label, const int *__restrict__ pix, int bgc, int iw, int ih) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (y >= ih) return;                       // whole wave leaves together (one wave = one row segment)
  const bool valid = x < iw;
  const int p = y * iw + x;
  const int v = valid ? pix[p] : 0;
  const int vl = __shfl_up(v, 1);
  const bool vlvalid = threadIdx.x > 0;      // lane 0 always starts a run; its left neighbour is handled by the merge phase
  const bool same = valid && vlvalid && vl == v;
  const unsigned long long starts = __ballot(!same);
  const unsigned long long upto = starts & ((2ull << threadIdx.x) - 1ull);
  const int start = 63 - __clzll((long long)upto);
  if (valid) label[p] = v == bgc ? -1 : y * iw + blockIdx.x * 64 + start;
}

// Phase 2: unions across rows and across 64-pixel segment borders.  A pixel only issues a union when no pixel of
// its run to the left/right is guaranteed to issue an equivalent one (see the case analysis in DESIGN.md).
__global__ __launch_bounds__(256) void k_label_merge(int *label, const int *__restrict__ pix, int bgc, int iw, int ih) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= iw || y >= ih) return;
  const int p = y * iw + x;
  const int v = pix[p];
  if (v == bgc) return;
  const bool wSame = x > 0 && pix[p - 1] == v;
  if (wSame && threadIdx.x == 0) uf_union(label, p, p - 1);
  if (y == 0) return;
  const bool nSame = pix[p - iw] == v;
  const bool nwSame = x > 0 && pix[p - iw - 1] == v;
  if (nSame) {
    if (!(wSame && nwSame)) uf_union(label, p, p - iw);
  } else {
    const bool neSame = x < iw - 1 && pix[p - iw + 1] == v;
    const bool eSame = x < iw - 1 && pix[p + 1] == v;
    if (nwSame && !wSame) uf_union(label, p, p - iw - 1);
    if (neSame && !eSame) uf_union(label, p, p - iw + 1);
  }
}

// Phase 3: path compression to the root
__global__ __launch_bounds__(256) void k_label_flatten(int *label, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int l = label[i];
    if (l >= 0) {
      const int r = uf_find(label, l);
      if (r != l) label[i] = r;
    }
  }
}

// oclimgutil.cl:641-649: out[label] += (int)(e*e*10000) for interior pixels with label > 0.  Lanes of a wave that
// share a label are summed with a ballot/shuffle loop first, so a big component costs one atomic per wave instead of
// one per pixel; zero contributions (most pixels) are skipped.  Integer addition: order independent.
__global__ __launch_bounds__(256) void k_calc_strength(int *out, const float *__restrict__ edge, const int *__restrict__ label, int iw, int ih) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  int l = -1, val = 0;
  if (x > 0 && y > 0 && x < iw - 1 && y < ih - 1) {
    const int p = y * iw + x;
    l = label[p];
    if (l > 0) { const float e = edge[p]; val = (int)(e * e * 10000.0f); }
  }
  bool todo = l > 0 && val != 0;
  while (__any(todo)) {
    unsigned long long m = __ballot(todo);
    const int leader = __ffsll((long long)m) - 1;
    const int ll = __shfl(l, leader);
    const bool mine = todo && l == ll;
    int sum = mine ? val : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if ((int)threadIdx.x == leader) atomicAdd(&out[ll], sum);
    if (mine) todo = false;
  }
}

// oclimgutil.cl:651-657
__global__ __launch_bounds__(256) void k_filter_strength(int *label, const int *__restrict__ str, int thre, int iw, int ih) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x <= 0 || y <= 0 || x >= iw - 1 || y >= ih - 1) return;
  const int p = y * iw + x;
  const int l = label[p];
  if (l <= 0 || str[l] < thre) label[p] = -1;
}

}  // namespace

namespace rdk {

void label8(hipStream_t s, int *label, const int *pix, int bgc, int iw, int ih) {
  hipLaunchKernelGGL(k_label_init, grid2(iw, ih), block2, 0, s, label, pix, bgc, iw, ih);
  hipLaunchKernelGGL(k_label_merge, grid2(iw, ih), block2, 0, s, label, pix, bgc, iw, ih);
  const int n = iw * ih;
  int g = cdiv(n, 256 * 4);
  hipLaunchKernelGGL(k_label_flatten, dim3(g < 1 ? 1 : g), dim3(256), 0, s, label, n);
}

void calc_strength(hipStream_t s, int *out, const float *edge, const int *label, int iw, int ih) {
  hipLaunchKernelGGL(k_calc_strength, grid2(iw, ih), block2, 0, s, out, edge, label, iw, ih);
}

void filter_strength(hipStream_t s, int *label, const int *str, int thre, int iw, int ih) {
  hipLaunchKernelGGL(k_filter_strength, grid2(iw, ih), block2, 0, s, label, str, thre, iw, ih);
}

}  // namespace rdk
