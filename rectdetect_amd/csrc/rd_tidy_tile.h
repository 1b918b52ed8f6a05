// rectdetect-mi355x: the rect-variant edge tidy of one 64 x ROWS tile (shared by the labelling tile kernel), on BIT ROWS.
#pragma once
#include "rd_device.h"

namespace rd {

// One row of a tile with its margin as 128 bits: bit b = column x0 - TD_M + b (72 of them in use).  The four 3x3 stencils of the
// tidy only ever ask "is the cell on" and "is its count 2", so a stencil on a whole row is a handful of word operations on the rows
// above, at and below it - about a hundred times fewer vector instructions than evaluating it cell by cell on bytes.
struct bitrow { unsigned long long lo, hi; };
__device__ __forceinline__ bitrow br(unsigned long long lo, unsigned long long hi) { bitrow r; r.lo = lo; r.hi = hi; return r; }
__device__ __forceinline__ bitrow operator&(bitrow a, bitrow b) { return br(a.lo & b.lo, a.hi & b.hi); }
__device__ __forceinline__ bitrow operator|(bitrow a, bitrow b) { return br(a.lo | b.lo, a.hi | b.hi); }
__device__ __forceinline__ bitrow operator~(bitrow a) { return br(~a.lo, ~a.hi); }
__device__ __forceinline__ bitrow br_west(bitrow a) { return br(a.lo << 1, (a.hi << 1) | (a.lo >> 63)); }      // every cell sees its WEST neighbour's bit
__device__ __forceinline__ bitrow br_east(bitrow a) { return br((a.lo >> 1) | (a.hi << 63), a.hi >> 1); }      // ... its EAST neighbour's
__device__ __forceinline__ int br_bit(bitrow a, int b) { return (int)(((b < 64 ? a.lo >> b : a.hi >> (b - 64))) & 1ull); }
// bits first..last (inclusive; any integers) of a 128-bit row
__device__ __forceinline__ bitrow br_range(int first, int last) {
  if (first < 0) first = 0;
  if (last > 127) last = 127;
  if (first > last) return br(0, 0);
  const unsigned long long lo = first < 64 ? ((~0ull << first) & (last >= 63 ? ~0ull : ((1ull << (last + 1)) - 1ull))) : 0ull;
  const int f2 = first < 64 ? 0 : first - 64, l2 = last - 64;
  const unsigned long long hi = l2 < 0 ? 0ull : ((~0ull << f2) & (l2 >= 63 ? ~0ull : ((1ull << (l2 + 1)) - 1ull)));
  return br(lo, hi);
}
// saturating count of the eight neighbours of every cell of row m (u above, d below): ge1 = at least one is on, ge2 = at least two
__device__ __forceinline__ void br_count8(bitrow u, bitrow m, bitrow d, bitrow &ge1, bitrow &ge2) {
  const bitrow in[8] = { br_west(u), u, br_east(u), br_west(m), br_east(m), br_west(d), d, br_east(d) };
  ge1 = br(0, 0); ge2 = br(0, 0);
#pragma unroll
  for (int k = 0; k < 8; k++) { ge2 = ge2 | (ge1 & in[k]); ge1 = ge1 | in[k]; }
}

// mask (NMS response > 0, rh:262-264) -> junction counts -> gap closing -> two thinning passes (rh:266-272; rc:67-135) for one
// 64 x ROWS tile: the four stencils need 4 cells of margin in total.
// Block of 256 threads (64 x 4), tid = ty * 64 + tx.  A, B: (ROWS + 8) * 72 bytes of LDS each, 16-byte aligned (they hold three bit
// planes each).  The results of the tile's own pixels are stored to mask0 (if given) / tidy (and zero_plane cleared, if given) and returned in
// outv[j] for row ty + 4 j, column tx (0 for pixels outside the frame).
#define TD_M 4
#define TD_P (64 + 2 * TD_M)
template <int ROWS>
__device__ __forceinline__ void rect_tidy_tile(uint8_t *A, uint8_t *B, int x0, int y0, int tid, const float *__restrict__ nms, int *__restrict__ mask0, int *__restrict__ tidy,
                                               int *__restrict__ zero_plane, int iw, int ih, int (&outv)[ROWS / 4]) {
  constexpr int R = ROWS + 2 * TD_M;
  static_assert(R * 3 * (int)sizeof(bitrow) <= R * TD_P, "three bit planes per byte buffer");
  bitrow *M = (bitrow *)A, *NZ = M + R, *E2 = NZ + R;          // the mask; count != 0; count == 2 (rc:67-95 keeps nothing else of the counts)
  bitrow *O = (bitrow *)B, *T0 = O + R;                        // after gap closing; after the first thinning pass
  const int lane = tid & 63, w = tid >> 6;
  // ---- the mask as bit rows: wave w takes rows w, w + 4, ...; lanes = columns -4..59, then lanes 0..7 = columns 60..67.
  //      All loads of a thread first (clamped addresses), then the ballots.
  {
    constexpr int RW = (R + 3) / 4;
    float fa[RW], fb[RW];
    bool oka[RW], okb[RW];
#pragma unroll
    for (int k = 0; k < RW; k++) {
      const int r = w + 4 * k, y = y0 - TD_M + r;
      const int xa = x0 - TD_M + lane, xb = x0 - TD_M + 64 + lane;
      const bool rowin = r < R && y >= 0 && y < ih;
      oka[k] = rowin && xa >= 0 && xa < iw;
      okb[k] = rowin && lane < 8 && xb < iw;
      fa[k] = nms[oka[k] ? y * iw + xa : 0];
      fb[k] = nms[okb[k] ? y * iw + xb : 0];
    }
#pragma unroll
    for (int k = 0; k < RW; k++) {
      const int r = w + 4 * k, y = y0 - TD_M + r;
      const bool va = oka[k] && fa[k] > 0.0f, vb = okb[k] && fb[k] > 0.0f;
      const unsigned long long ba = __ballot(va), bb = __ballot(vb);
      if (r < R && lane == 0) M[r] = br(ba, bb);
      // the tile's own cells of the mask plane: columns 0..59 from the first load, 60..63 from the second
      const int ta = lane - TD_M, tb = lane + 64 - TD_M, tr = r - TD_M;
      if (mask0 != nullptr && tr >= 0 && tr < ROWS) {      // (the mask as an int plane: the operator's output; the frame path keeps it as these bit rows only)
        if (oka[k] && ta >= 0) mask0[y * iw + x0 + ta] = va ? 1 : 0;
        if (okb[k] && tb < 64) mask0[y * iw + x0 + tb] = vb ? 1 : 0;
      }
    }
  }
  __syncthreads();
  const int r = tid;                                   // (one thread per row for the stencils: R <= 64 rows, all in wave 0)
  const int y = y0 - TD_M + r;
  // cells where the stencils apply: x in [k, iw-1-k] (bit = x - x0 + TD_M), y in [k, ih-1-k]
  const bitrow in1 = (y >= 1 && y <= ih - 2) ? br_range(1 - x0 + TD_M, iw - 2 - x0 + TD_M) : br(0, 0);
  const bitrow in2 = (y >= 2 && y <= ih - 3) ? br_range(2 - x0 + TD_M, iw - 3 - x0 + TD_M) : br(0, 0);
  // checkerboard of this row: bit b is column x0 - TD_M + b; (x + y) even
  const unsigned long long even = (((x0 - TD_M + y) & 1) == 0) ? 0x5555555555555555ull : 0xaaaaaaaaaaaaaaaaull;
  if (r >= 1 && r < R - 1) {   // rc:67-95: on-pixels with at least one on-neighbour keep a count (!= 0); count 2 = exactly one neighbour
    bitrow ge1, ge2;
    br_count8(M[r - 1], M[r], M[r + 1], ge1, ge2);
    const bitrow on = M[r] & in1;
    NZ[r] = on & ge1;
    E2[r] = on & ge1 & ~ge2;
  } else if (r < R) { NZ[r] = br(0, 0); E2[r] = br(0, 0); }
  __syncthreads();
  if (r >= 2 && r < R - 2) {   // rc:97-121: on-pixels stay; 1-px gaps next to a curve end (count == 2) are closed by ten patterns
    const bitrow zn = NZ[r - 1], zm = NZ[r], zs = NZ[r + 1], en = E2[r - 1], em = E2[r], es = E2[r + 1];
    const bitrow pat = (br_west(em) & br_east(zm)) | (br_west(zm) & br_east(em)) | (en & zs) | (zn & es) |
                       (br_west(en) & br_east(es)) | (br_east(en) & br_west(es)) |
                       (br_east(em) & br_west(es)) | (br_west(em) & br_east(es)) | (br_east(en) & es) | (br_west(en) & es);
    O[r] = in2 & (zm | pat);
  } else if (r < R) O[r] = br(0, 0);
  __syncthreads();
  if (r >= 3 && r < R - 3) {   // rc:123-135: checkerboard thinning, parity 0: a pixel with an orthogonal L-shaped pair of on-neighbours goes
    const bitrow o = O[r];
    const bitrow kill = in1 & br(even, even) & (O[r - 1] | O[r + 1]) & (br_west(o) | br_east(o));
    T0[r] = o & ~kill;
  } else if (r < R) T0[r] = br(0, 0);
  __syncthreads();
  bitrow *T1 = E2;             // (free by now) parity 1, the tile's rows only
  if (r >= TD_M && r < R - TD_M) {
    const bitrow o = T0[r];
    const bitrow kill = in1 & br(~even, ~even) & (T0[r - 1] | T0[r + 1]) & (br_west(o) | br_east(o));
    T1[r] = o & ~kill;
  }
  __syncthreads();
  const int tx = lane, x = x0 + tx;
#pragma unroll
  for (int j = 0; j < ROWS / 4; j++) {
    const int tr = w + 4 * j, yy = y0 + tr;
    int v = 0;
    if (x < iw && yy < ih) {
      v = br_bit(T1[tr + TD_M], tx + TD_M);
      tidy[yy * iw + x] = v;
      if (zero_plane) zero_plane[yy * iw + x] = 0;
    }
    outv[j] = v;
  }
}

}  // namespace rd
