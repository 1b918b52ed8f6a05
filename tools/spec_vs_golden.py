#!/usr/bin/env python3
"""CPU study (no GPU): the oracle in a region mode (1 = the spec the HIP path reproduces: merge with concurrent work-items until nothing
changes, absorption in serial raster order) against a golden stream made by the reference in raster order (tests/golden/<name>.npz): which
frames' rectangle sets differ.  usage: spec_vs_golden.py <name> [mode] [max frames]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rectdetect_amd as ra
from rectdetect_amd import synth
from tests import helpers
name = sys.argv[1]; mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1; nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 10**9
g = np.load(os.path.join(helpers.GOLDEN, name + ".npz"))
iw, ih, tan, seed, nframes = int(g["iw"]), int(g["ih"]), float(g["tan_aov"]), int(g["seed"]), int(g["nframes"])
orc = helpers.OracleRect(iw, ih, mode)
canon = lambda rs: rs[np.lexsort(np.rint(rs["c2"]).reshape(len(rs), 8).T[::-1])] if len(rs) else rs
exact, bad = 0, []
for t in range(min(nframes, nmax)):
    orc.frame(synth.frame(seed, iw, ih, t))
    assert helpers.segments_equal(orc.segments(), g[f"f{t}_segments"]), t
    mine = ra.postprocess_planes(orc.segments(), orc.plane("boundary"), orc.plane("table"), iw, ih, tan)
    ref = g[f"f{t}_rects"]
    same = len(mine) == len(ref) and helpers.rects_equal(canon(mine), canon(ref))
    exact += same
    if not same:
        bad.append(t)
        print(name, "mode", mode, "frame", t, "differs from the raster-order list:", len(mine), "rectangles against", len(ref), "launches", orc.rounds()[0], flush=True)
print(name, "mode", mode, ": rectangle sets identical to the raster-order reference on", exact, "of", min(nframes, nmax), "frames; not identical:", bad, flush=True)
