#!/usr/bin/env python3
"""GPU box: the frames of a golden stream (tests/golden/<name>.npz, the reference in raster order) whose rectangle SET the HIP path does not
reproduce bit for bit - the frames tools/make_golden_stream_orders.py then runs the reference on under other legal work-item orders.
usage: stream_mismatch.py <name> [slots]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rectdetect_amd as ra
from rectdetect_amd import synth
from tests import helpers
name = sys.argv[1]; slots = int(sys.argv[2]) if len(sys.argv) > 2 else 16
g = np.load(os.path.join(helpers.GOLDEN, name + ".npz"))
iw, ih, nframes, tan, seed = int(g["iw"]), int(g["ih"]), int(g["nframes"]), float(g["tan_aov"]), int(g["seed"])
det = ra.Detector(iw, ih, nslots=slots, nworkers=1)
got, infl = [], 0
for t in range(nframes):
    if infl == slots:
        got.append((det.poll(tan), det.last_segments())); infl -= 1
    det.enqueue(synth.frame(seed, iw, ih, t)); infl += 1
while infl:
    got.append((det.poll(tan), det.last_segments())); infl -= 1
canon = lambda rs: rs[np.lexsort(np.rint(rs["c2"]).reshape(len(rs), 8).T[::-1])] if len(rs) else rs
bad = []
for t, (rects, segs) in enumerate(got):
    assert helpers.segments_equal(segs, g[f"f{t}_segments"]), t
    ref = g[f"f{t}_rects"]
    if not (len(rects) == len(ref) and helpers.rects_equal(canon(rects), canon(ref))):
        bad.append(t); print(name, "frame", t, ":", len(rects), "rectangles, reference", len(ref))
print(name, ": segment lists identical on all", nframes, "frames; rectangle sets identical on", nframes - len(bad), "; not identical:", bad)
det.close()
