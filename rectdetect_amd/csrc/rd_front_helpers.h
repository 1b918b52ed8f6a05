// rectdetect-mi355x: device helpers shared by the front-end kernels (rd_k_front.hip) and the fused gradient / suppression tile kernel
// (rd_k_nms.hip, a translation unit of its own because it is compiled without the SLP vectoriser: packed fp32 operations cost more than the two
// plain ones they replace in that kernel - profiles/NOTES_r04.md).
#pragma once
#include "rd_device.h"

namespace {
using namespace rd;

// iu:346-352 (5x5 kernel; double literals narrowed to float) and iu:395-420
__device__ __forceinline__ float v5c(int i) {
  const float t[25] = {
    (float)-4.667, (float)-4.083, (float)0.000, (float)4.083, (float)4.667,
    (float)-10.024, (float)-0.963, (float)0.000, (float)0.963, (float)10.024,
    (float)-14.120, (float)3.622, (float)0.000, (float)-3.622, (float)14.120,
    (float)-10.024, (float)-0.963, (float)0.000, (float)0.963, (float)10.024,
    (float)-4.667, (float)-4.083, (float)0.000, (float)4.083, (float)4.667,
  };
  return t[i];
}

__device__ __forceinline__ float ep_strength(const float (&n)[3], const float (&s)[3], const float (&w)[3], const float (&e)[3],
                                             const float (&nw)[3], const float (&ne)[3], const float (&sw)[3], const float (&se)[3]) {
  float sum[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    float t = n[c] + w[c] - s[c] - e[c];
    float acc = 0;
    acc += (nw[c] - se[c]) * t;
    t = n[c] - w[c] + e[c] - s[c];
    acc += (ne[c] - sw[c]) * t;
    sum[c] = fmaxf(0.0f, acc);
  }
  const float tot = sum[0] + sum[1] + sum[2];
  return tot > 0 ? sqrtf(tot) : 0.0f;
}

#define TT_PITCH 72
__device__ __forceinline__ float bicubic_lds(const float *t, float x, float y, int x0, int y0) {
  const int ix = (int)x, iy = (int)y;
  const float fx = x - ix, fy = y - iy;
  const float *q = t + (iy - 1 - (y0 - 3)) * TT_PITCH + (ix - 1 - (x0 - 3));
  float r[4];
#pragma unroll
  for (int k = 0; k < 4; k++) r[k] = cubic1(q[k * TT_PITCH], q[k * TT_PITCH + 1], q[k * TT_PITCH + 2], q[k * TT_PITCH + 3], fx);
  return cubic1(r[0], r[1], r[2], r[3], fy);
}

}  // namespace
