"""Synthetic BGR frames - numpy twin of csrc/rd_synth.c, bit-identical output.

Used for golden fixtures, tests and bench input; replaces the reference's OpenCV frame sources
(rect.cpp:66-74, vidrect.cpp:55-99), which are unavailable in this environment.
"""
import numpy as np

SEED0 = 0x5EED0000
_M64 = (1 << 64) - 1


class _Rng:
    def __init__(self, s):
        self.s = s & _M64 or 1

    def next(self):
        x = self.s
        x ^= x >> 12
        x ^= (x << 25) & _M64
        x ^= x >> 27
        self.s = x
        return ((x * 0x2545F4914F6CDD1D) & _M64) >> 32

    def randint(self, lo, hi):
        return lo + self.next() % (hi - lo + 1)


def num_quads(iw, ih):
    return max(3, (12 * iw * ih + 1036800) // 2073600)


def quad(seed, iw, ih, t, k):
    """Corners [(x,y)]*4 and colour (b,g,r) of quad k at time t."""
    r = _Rng(seed ^ ((0x9E3779B97F4A7C15 * (k + 1)) & _M64))
    cx, cy = r.randint(0, iw - 1), r.randint(0, ih - 1)
    a, b = r.randint(ih // 16, ih // 5), r.randint(ih // 16, ih // 5)
    j = min(a, b) // 3
    sx, sy = (-1, 1, 1, -1), (-1, -1, 1, 1)
    off = []
    for i in range(4):
        ox = sx[i] * a + r.randint(-j, j)
        oy = sy[i] * b + r.randint(-j, j)
        off.append((ox, oy))
    col = [r.randint(60, 250) for _ in range(3)]
    if max(col) < 80:
        col[k % 3] += 40
    vx, vy = r.randint(-3, 3), r.randint(-3, 3)
    px, py = (cx + vx * t) % iw, (cy + vy * t) % ih
    return [(px + ox, py + oy) for ox, oy in off], tuple(col)


def _noise_hash(x, y, t, c, seed32):
    u = np.uint32
    with np.errstate(over="ignore"):
        h = u(seed32) ^ (x * u(0x9E3779B1)) ^ (y * u(0x85EBCA77)) ^ u((t * 0xC2B2AE3D) & 0xFFFFFFFF) ^ u((c * 0x27D4EB2F) & 0xFFFFFFFF)
        h ^= h >> u(15)
        h *= u(0x2C1B3C6D)
        h ^= h >> u(12)
        h *= u(0x297A2D39)
        h ^= h >> u(15)
    return h


def frame(seed, iw, ih, t=0, noise=True):
    """uint8 array (ih, iw, 3), BGR, row stride 3*iw."""
    img = np.full((ih, iw, 3), 40, np.uint8)
    for k in range(num_quads(iw, ih)):
        q, col = quad(seed, iw, ih, t, k)
        xs, ys = [p[0] for p in q], [p[1] for p in q]
        x0, x1 = max(min(xs), 0), min(max(xs), iw - 1)
        y0, y1 = max(min(ys), 0), min(max(ys), ih - 1)
        if x0 > x1 or y0 > y1:
            continue
        yy, xx = np.mgrid[y0:y1 + 1, x0:x1 + 1].astype(np.int64)
        inside = np.ones(yy.shape, bool)
        for i in range(4):
            ax, ay = q[i]
            bx, by = q[(i + 1) & 3]
            inside &= ((bx - ax) * (yy - ay) - (by - ay) * (xx - ax)) >= 0
        img[y0:y1 + 1, x0:x1 + 1][inside] = col
    if noise:
        s32 = (seed ^ (seed >> 32)) & 0xFFFFFFFF
        yy, xx = np.mgrid[0:ih, 0:iw].astype(np.uint32)
        out = img.astype(np.int32)
        for c in range(3):
            out[:, :, c] += (_noise_hash(xx, yy, t, c, s32) >> np.uint32(29)).astype(np.int32) - 4
        img = np.clip(out, 0, 255).astype(np.uint8)
    return img


def hard_frame(kind, seed, iw, ih):
    """Inputs that are much busier than the stream generator's flat quads (tests only; numpy's PCG64 streams are stable):
    'tiles'  random colour tiles of 8..24 px (thousands of regions and junctions),
    'noise'  independent uniform noise per pixel and channel,
    'waves'  smooth sinusoidal colour fields with a few dark-rimmed bright rectangles on top,
    'bars'   horizontal and vertical bars of varying width on a gradient."""
    rng = np.random.default_rng([seed, iw, ih])
    if kind == "tiles":
        t = int(rng.integers(8, 25))
        tiles = rng.integers(0, 256, ((ih + t - 1) // t, (iw + t - 1) // t, 3), dtype=np.uint8)
        return np.ascontiguousarray(np.repeat(np.repeat(tiles, t, 0), t, 1)[:ih, :iw])
    if kind == "noise":
        return rng.integers(0, 256, (ih, iw, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:ih, 0:iw].astype(np.float64)
    if kind == "waves":
        img = np.stack([127 + 100 * np.sin(xx * rng.uniform(0.01, 0.05) + yy * rng.uniform(0.0, 0.03) + rng.uniform(0, 6)) for _ in range(3)], -1)
        for _ in range(5):
            x0, y0 = int(rng.integers(0, iw - 80)), int(rng.integers(0, ih - 60))
            w, h = int(rng.integers(40, iw // 3)), int(rng.integers(30, ih // 3))
            img[y0:y0 + h, x0:x0 + w] = 20
            img[y0 + 3:y0 + h - 3, x0 + 3:x0 + w - 3] = rng.integers(150, 256, 3)
        return np.clip(img, 0, 255).astype(np.uint8)
    if kind == "bars":
        img = np.stack([xx * 255 / iw, yy * 255 / ih, (xx + yy) * 255 / (iw + ih)], -1)
        for _ in range(12):
            c = rng.integers(0, 256, 3)
            if rng.integers(0, 2):
                x0 = int(rng.integers(0, iw - 4)); img[:, x0:x0 + int(rng.integers(2, 30))] = c
            else:
                y0 = int(rng.integers(0, ih - 4)); img[y0:y0 + int(rng.integers(2, 30)), :] = c
        return np.clip(img, 0, 255).astype(np.uint8)
    raise ValueError(kind)
