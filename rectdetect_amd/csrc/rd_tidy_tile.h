// rectdetect-mi355x: the rect-variant edge tidy of one 64 x ROWS tile in LDS (shared by the labelling tile kernel).
#pragma once
#include "rd_device.h"

namespace rd {

// mask (NMS response > 0, rh:262-264) -> junction counts -> gap closing -> two thinning passes (rh:266-272) for one
// 64 x ROWS tile in LDS: the four 3x3 stencils need 4 cells of halo in total, every intermediate is a byte.
// Block of 256 threads (64 x 4), tid = ty * 64 + tx.  A, B: (ROWS + 8) * 72 bytes of LDS each.  The results of the tile's own
// pixels are stored to mask0 / tidy (and zero_plane cleared, if given) and returned in outv[j] for row ty + 4 j, column tx
// (0 for pixels outside the frame) - the thread mapping of the last pass.
#define TD_M 4
#define TD_P (64 + 2 * TD_M)
template <int ROWS>
__device__ __forceinline__ void rect_tidy_tile(uint8_t *A, uint8_t *B, int x0, int y0, int tid, const float *__restrict__ nms, int *__restrict__ mask0, int *__restrict__ tidy,
                                               int *__restrict__ zero_plane, int iw, int ih, int (&outv)[ROWS / 4]) {
  // region with margin m around the tile, cell t -> (r, c) tile coordinates, (x, y) frame coordinates, i = LDS index
#define TD_FOR(m) for (int t = tid; t < (ROWS + 2 * (m)) * (64 + 2 * (m)); t += 256)
#define TD_CELL(m) const int r = t / (64 + 2 * (m)) - (m), c = t % (64 + 2 * (m)) - (m); const int x = x0 + c, y = y0 + r; const int i = (r + TD_M) * TD_P + c + TD_M; const bool in_img = x >= 0 && x < iw && y >= 0 && y < ih
  stage_cells<(ROWS + 8) * (64 + 8), 256>(tid, nms,
    [&](int t, int &a) { TD_CELL(4); (void)i; a = y * iw + x; return in_img; },
    [&](int t, bool ok, float f) {
      TD_CELL(4);
      const uint8_t v = (ok && f > 0.0f) ? 1 : 0;
      if (in_img && r >= 0 && r < ROWS && c >= 0 && c < 64) mask0[y * iw + x] = v;
      A[i] = v;
    });
  __syncthreads();
  TD_FOR(3) {   // rc:67-95 (count of on-pixels in the 3x3 block, isolated pixels -> 0)
    TD_CELL(3);
    uint8_t v = 0;
    if (in_img && x > 0 && y > 0 && x < iw - 1 && y < ih - 1 && A[i] != 0) {
      int count = 1;
#pragma unroll
      for (int k = 0; k < 8; k++) count += A[i + nbr_dx(k) + nbr_dy(k) * TD_P] != 0;
      v = count == 1 ? 0 : count;
    }
    B[i] = v;
  }
  __syncthreads();
  TD_FOR(2) {   // rc:97-121: on-pixels stay, 1-px gaps next to a curve end (count == 2) are closed by ten patterns; 2-px ring -> 0
    TD_CELL(2);
    uint8_t o = 0;
    if (in_img && x > 1 && y > 1 && x < iw - 2 && y < ih - 2) {
      if (B[i] != 0) o = 1;
      else {
        const int w = B[i - 1], e = B[i + 1], n = B[i - TD_P], sdn = B[i + TD_P];
        const int nw = B[i - TD_P - 1], ne = B[i - TD_P + 1], sw = B[i + TD_P - 1], se = B[i + TD_P + 1];
        if ((w == 2 && e != 0) || (w != 0 && e == 2) || (n == 2 && sdn != 0) || (n != 0 && sdn == 2) || (nw == 2 && se == 2) || (ne == 2 && sw == 2) ||
            (e == 2 && sw == 2) || (w == 2 && se == 2) || (ne == 2 && sdn == 2) || (nw == 2 && sdn == 2)) o = 1;
      }
    }
    A[i] = o;
  }
  __syncthreads();
  TD_FOR(1) {   // rc:123-135: checkerboard thinning - a pixel of this parity with an orthogonal L-shaped pair of on-neighbours goes; parity 0
    TD_CELL(1);
    uint8_t v = in_img ? A[i] : 0;
    if (in_img && x > 0 && y > 0 && x < iw - 1 && y < ih - 1 && ((x + y) & 1) == 0) {
      if ((A[i - TD_P] != 0 || A[i + TD_P] != 0) && (A[i - 1] != 0 || A[i + 1] != 0)) v = 0;
    }
    B[i] = v;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < ROWS / 4; j++) {   // parity 1; cell t = tid + 256 j is row ty + 4 j, column tx
    const int t = tid + 256 * j;
    TD_CELL(0);
    int v = 0;
    if (in_img) {
      v = B[i];
      if (x > 0 && y > 0 && x < iw - 1 && y < ih - 1 && ((x + y) & 1) == 1) {
        if ((B[i - TD_P] != 0 || B[i + TD_P] != 0) && (B[i - 1] != 0 || B[i + 1] != 0)) v = 0;
      }
      tidy[y * iw + x] = v;
      if (zero_plane) zero_plane[y * iw + x] = 0;
    }
    outv[j] = v;
  }
#undef TD_FOR
#undef TD_CELL
}

}  // namespace rd
