"""Exhaustive check of the float formulation of floor(s / w) used by the edge-stopped blur (rd_k_rect.hip, div_small_f):
floor(fma(float(s), 1.0f / float(w), 0.5f * (1.0f / float(w)))) == s // w for every 0 <= s <= 40950 (ten 12-bit samples), 1 <= w <= 10,
with an IEEE single-precision fused multiply-add and a correctly rounded divide (what the kernel uses).  The fma is emulated exactly:
the product of a 16-bit integer and a 24-bit significand plus the half-reciprocal has fewer than 53 significant bits, so the
double-precision expression is exact and the conversion to single precision rounds once."""
import numpy as np

s = np.arange(0, 40951, dtype=np.float64)
for w in range(1, 11):
    rw = np.float32(1.0) / np.float32(w)
    half = np.float32(0.5) * rw
    exact = s * np.float64(rw) + np.float64(half)
    got = np.floor(exact.astype(np.float32)).astype(np.int64)
    assert np.array_equal(got, np.arange(0, 40951) // w), w
print("div_small_f exact for s <= 40950, w <= 10")
