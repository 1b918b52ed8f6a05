"""Exhaustive check of the integer formulation of floor(s / w) used by the edge-stopped blur (rd_k_rect.hip, div_small_m):
(s * ceil(2^19 / w)) >> 19 == s // w for every sum s of w samples of a 12-bit field (0 <= s <= 4095 w), 1 <= w <= 10, with the product below 2^32
and both factors below 2^24 (v_mul_u32_u24)."""
for w in range(1, 11):
    m = -(-(1 << 19) // w)
    assert m < (1 << 24) and 4095 * w * m < (1 << 32)
    for s in range(0, 4095 * w + 1):
        assert (s * m) >> 19 == s // w, (w, s)
print("div_small_m exact for s <= 4095 w, w <= 10")
