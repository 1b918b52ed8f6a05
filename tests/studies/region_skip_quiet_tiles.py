"""Study (not a test; profiles/NOTES_r05.md): how many 64x48 tiles of the region merge could skip a launch, with near changes tracked per tile (stamps) and far
reads checked word by word (the marks) - simulated with numpy on the oracle planes of the bench stream.  python tests/studies/region_skip_quiet_tiles.py [frames before]"""
import os, sys, numpy as np, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import helpers
from rectdetect_amd import synth
iw, ih = 1920, 1080
N = iw * ih
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 1
o = helpers.OracleRect(iw, ih, helpers.REGION_SPEC)
for t in range(nf):
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
SIZES = [(64, 48), (64, 24), (64, 12), (64, 6)]
def units(TW, TH):
    gx, gy = (iw + TW - 1) // TW, (ih + TH - 1) // TH
    return gx, gy
def unit_of(q, TW, TH, gx):
    return (q // iw) // TH * gx + (q % iw) // TW
prev_changed = None
tot = {s: [0, 0, 0] for s in SIZES}
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
    R = np.stack(reads, 1)
    if prev_changed is not None:
        chg = np.zeros(N, bool); chg[prev_changed] = True
        # pixel must run: any near word or far word changed last round
        near = chg[p0] | chg[p0 - iw] | chg[p0 - 1] | chg[p0 + 1] | chg[p0 + iw]
        far = chg[R].any(1)
        must = near | far
        line = "round %2d: changed(prev) %8d  pixels that must run %8d (near %8d far-only %8d)" % (r, len(prev_changed), must.sum(), near.sum(), (far & ~near).sum())
        for (TW, TH) in SIZES:
            gx, gy = units(TW, TH)
            nu = gx * gy
            u0 = unit_of(p0, TW, TH, gx)
            exact = np.zeros(nu, bool); exact[np.unique(u0[must])] = True
            # practical: near by unit C bits (self + 4 adjacent units), far exact words
            cu = np.zeros(nu, bool); cu[np.unique(unit_of(prev_changed, TW, TH, gx))] = True
            c2 = cu.reshape(gy, gx)
            nearu = c2.copy(); nearu[1:] |= c2[:-1]; nearu[:-1] |= c2[1:]; nearu[:, 1:] |= c2[:, :-1]; nearu[:, :-1] |= c2[:, 1:]
            faru = np.zeros(nu, bool); faru[np.unique(u0[far])] = True
            prac = nearu.ravel() | faru
            line += " | %dx%d: exact %5d prac %5d of %5d" % (TW, TH, exact.sum(), prac.sum(), nu)
            tot[(TW, TH)][0] += exact.sum() / nu; tot[(TW, TH)][1] += prac.sum() / nu; tot[(TW, TH)][2] += 1
        print(line, flush=True)
        if r >= 3 and r <= 12:
            # far list sizes for 64x48 units: distinct far words outside the unit's own near units
            TW, TH = 64, 48
            gx, gy = units(TW, TH)
            u0 = unit_of(p0, TW, TH, gx)
            ru = unit_of(R, TW, TH, gx)
            ux, uy = u0 % gx, u0 // gx
            rx, ry = ru % gx, ru // gx
            outside = (np.abs(rx - ux[:, None]) + np.abs(ry - uy[:, None])) > 1
            key = (u0[:, None].astype(np.int64) * N + R)[outside]
            uk = np.unique(key)
            cnt = np.bincount((uk // N).astype(np.int64), minlength=gx * gy)
            print("      64x48 far words per unit: mean %.1f median %d p90 %d max %d; units over 32: %d, over 64: %d" % (cnt.mean(), np.median(cnt), np.percentile(cnt, 90), cnt.max(), (cnt > 32).sum(), (cnt > 64).sum()), flush=True)
    ch = g != og
    nxt = lab.copy()
    np.minimum.at(nxt, og[ch], g[ch])
    np.minimum.at(nxt, p0[ch], g[ch])
    changed = np.nonzero(nxt != lab)[0]
    prev_changed = changed
    lab = nxt
    if len(changed) == 0:
        print("round %d changed nothing" % r); break
for s in SIZES: print(s, "round-equivalents exact %.2f prac %.2f of %d" % tuple(tot[s]))
