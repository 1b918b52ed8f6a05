"""Study (not a test; profiles/NOTES_r04.md): which 64x48 tiles of the region merge repeat their previous round - simulated with numpy on the planes the CPU oracle produces
for frame 0 of the bench stream.  python tests/studies/region_tile_activity.py [frames to run before the one studied]"""
import os
import sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import helpers
from rectdetect_amd import synth
iw, ih = 1920, 1080
N = iw * ih
o = helpers.OracleRect(iw, ih, helpers.REGION_SPEC)
for t in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    o.frame(synth.frame(synth.SEED0, iw, ih, t))
pix = o.plane("quant").view(np.int32).reshape(ih, iw); mask = o.plane("mergemask").reshape(ih, iw); edge = o.plane("label1").reshape(ih, iw)
print("oracle rounds", o.rounds())
lab = np.arange(N, dtype=np.int64).reshape(ih, iw)
up = np.zeros((ih, iw), bool); up[1:] = pix[1:] == pix[:-1]
left = np.zeros((ih, iw), bool); left[:, 1:] = pix[:, 1:] == pix[:, :-1]
lab = np.where(up, lab - iw, np.where(left, lab - 1, lab)).ravel()
ys, xs = np.mgrid[1:ih-1, 1:iw-1]
p0 = (ys * iw + xs).ravel()
pf, mf, ef = pix.ravel(), mask.ravel() != 0, edge.ravel()
def allow(p1, ep):
    return ((pf[p0] == pf[p1]) | mf[p0]) & (ef[ep] <= 0)
A = [(p0 - iw, allow(p0 - iw, p0)), (p0 - 1, allow(p0 - 1, p0)), (p0 + 1, allow(p0 + 1, p0 + 1)), (p0 + iw, allow(p0 + iw, p0 + iw))]
TW, TH = 64, 48
gx, gy = (iw + TW - 1) // TW, (ih + TH - 1) // TH
def tile_of(q):
    return (q // iw) // TH * gx + (q % iw) // TW
SH = 15
t0 = tile_of(p0)
ntile = gx * gy
# near neighbourhood of tiles (4-neighbours + self)
nb = [[] for _ in range(ntile)]
for ty in range(gy):
    for tx in range(gx):
        T = ty * gx + tx
        for dy, dx in ((0,0),(1,0),(-1,0),(0,1),(0,-1)):
            if 0 <= ty+dy < gy and 0 <= tx+dx < gx: nb[T].append((ty+dy)*gx+tx+dx)
far_chunks = np.zeros(ntile, np.uint64)      # stored read set (chunks), from the last round the tile ran
far_tiles = [set() for _ in range(ntile)]
active_a = np.ones(ntile, bool); active_b = np.ones(ntile, bool)
for r in range(40):
    og = lab[p0]
    g = og.copy()
    for p1, al in A:
        s = lab[p1]
        g = np.where((s < g) & al, s, g)
    reads = []
    for j in range(8):
        reads.append(g.copy())
        g = lab[g]
    ch = g != og
    nxt = lab.copy()
    np.minimum.at(nxt, og[ch], g[ch])
    np.minimum.at(nxt, p0[ch], g[ch])
    changed = np.nonzero(nxt != lab)[0]
    ctiles = np.zeros(ntile, bool); ctiles[np.unique(tile_of(changed))] = True
    cch = np.uint64(0)
    for c in np.unique(changed >> SH): cch |= np.uint64(1) << np.uint64(c)
    # tiles that had work this round (some pixel todo) - for reference
    work = np.zeros(ntile, bool); work[np.unique(t0[ch])] = True
    # read sets of this round (for tiles that ran): far reads = all jump reads
    R = np.stack(reads, 1)
    # criterion a: near by tile, far by chunk;  b: near by tile, far by tile
    rc = (R >> SH).astype(np.uint64)
    bits = np.bitwise_or.reduce(np.uint64(1) << rc, axis=1)
    fc = np.zeros(ntile, np.uint64); np.bitwise_or.at(fc, t0, bits)
    ft = [set() for _ in range(ntile)]
    RT = tile_of(R)
    for j in range(8):
        u = np.unique(t0 * 100000 + RT[:, j])
        for v in u: ft[v // 100000].add(int(v % 100000))
    print("round %2d: changed words %8d  tiles with a changed word %4d  tiles with a proposing pixel %4d  chunks changed %2d | ran: a %4d b %4d of %d" % (r, len(changed), ctiles.sum(), work.sum(), bin(int(cch)).count("1"), active_a.sum(), active_b.sum(), ntile), flush=True)
    # a tile that ran stores its read set; one that was skipped keeps it (simulate: all ran here since we compute everything; the stored set of a skipped tile equals the fresh one when the criterion is exact)
    na = np.zeros(ntile, bool); nbb = np.zeros(ntile, bool)
    ct_idx = set(np.nonzero(ctiles)[0].tolist())
    for T in range(ntile):
        near = any(ctiles[q] for q in nb[T])
        na[T] = near or bool(fc[T] & cch)
        nbb[T] = near or bool(ft[T] & ct_idx)
    active_a, active_b = na, nbb
    lab = nxt
    if len(changed) == 0: break
