"""Study (not a test; profiles/NOTES_r05.md): how often the eight jumps of launch 1 of the region merge leave a window of the label plane around the tile (rows above, columns left / right),
simulated with numpy on the oracle planes of the bench stream.  python tests/studies/region_launch1_window.py [frames before]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import helpers
from rectdetect_amd import synth
iw, ih = 1920, 1080
N = iw * ih
o = helpers.OracleRect(iw, ih, helpers.REGION_SPEC)
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for t in range(nf): o.frame(synth.frame(synth.SEED0, iw, ih, t))
pix = o.plane("quant").view(np.int32).reshape(ih, iw); mask = o.plane("mergemask").reshape(ih, iw); edge = o.plane("label1").reshape(ih, iw)
up = np.zeros((ih, iw), bool); up[1:] = pix[1:] == pix[:-1]
left = np.zeros((ih, iw), bool); left[:, 1:] = pix[:, 1:] == pix[:, :-1]
lab = np.arange(N, dtype=np.int64).reshape(ih, iw)
lab = np.where(up, lab - iw, np.where(left, lab - 1, lab)).ravel()
ys, xs = np.mgrid[1:ih-1, 1:iw-1]
p0 = (ys * iw + xs).ravel()
pf, mf, ef = pix.ravel(), mask.ravel() != 0, edge.ravel()
def allow(p1, ep):
    return ((pf[p0] == pf[p1]) | mf[p0]) & (ef[ep] <= 0)
A = [(p0 - iw, allow(p0 - iw, p0)), (p0 - 1, allow(p0 - 1, p0)), (p0 + 1, allow(p0 + 1, p0 + 1)), (p0 + iw, allow(p0 + iw, p0 + iw))]
def step():
    global lab
    og = lab[p0]; g = og.copy()
    for p1, al in A:
        s = lab[p1]; g = np.where((s < g) & al, s, g)
    chain = [g.copy()]
    for j in range(8):
        g = lab[g]; chain.append(g.copy())
    ch = g != og
    nxt = lab.copy()
    np.minimum.at(nxt, og[ch], g[ch]); np.minimum.at(nxt, p0[ch], g[ch])
    return chain, nxt
chain, nxt = step(); lab = nxt      # round 0 (k_region_init)
chain, nxt = step()                 # round 1
TH = 32
y0 = (p0 // iw) // TH * TH; x0 = (p0 % iw) // 64 * 64
for HA, HL, HR in [(96, 8, 8), (96, 16, 8), (128, 8, 8), (64, 8, 8), (80, 8, 4), (96, 4, 4)]:
    out_any = np.zeros(len(p0), bool); first_out = np.full(len(p0), 9)
    for j, c in enumerate(chain):     # chain[0] = m, chain[j] = after j jumps; reads happen at chain[0..7]
        cy, cx = c // iw, c % iw
        outside = (cy < y0 - HA) | (cx < x0 - HL) | (cx > x0 + 63 + HR) | (cy > y0 + TH)
        if j < 8:
            newly = outside & ~out_any
            first_out[newly] = j
            out_any |= outside
    print("window HA %d HL %d HR %d: pixels whose chain leaves it: %.3f %%; gathers that would go to memory (from the exit on): %.3f %% of all" % (HA, HL, HR, 100 * out_any.mean(), 100 * (8 - first_out[out_any]).sum() / (8.0 * len(p0))))
