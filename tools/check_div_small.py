"""Exhaustive check of the float formulation of floor(s / w) used by the edge-stopped blur (rd_k_rect.hip, div_small_f):
floor((float(s) + 0.5f) * (1.0f / float(w))) == s // w for every 0 <= s <= 40950 (ten 12-bit samples), 1 <= w <= 10,
with IEEE single-precision add, multiply and divide (what the kernel uses: no contraction, correctly rounded divide)."""
import numpy as np

s = np.arange(0, 40951, dtype=np.float32)
for w in range(1, 11):
    rw = np.float32(1.0) / np.float32(w)
    got = np.floor((s + np.float32(0.5)) * rw).astype(np.int64)
    assert np.array_equal(got, np.arange(0, 40951) // w), w
print("div_small_f exact for s <= 40950, w <= 10")
