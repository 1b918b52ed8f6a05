#!/usr/bin/env python3
"""Frames of the long golden streams on which this build's rectangle list differs from the reference's raster-order list
(tests/golden/stream_*.npz): prints them and saves this build's lists to gpurun_out/stream_mismatch.npz.  GPU box only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rectdetect_amd as ra  # noqa: E402
from rectdetect_amd import synth  # noqa: E402
from tests import helpers  # noqa: E402


def canon(rs):
    return rs[np.lexsort(np.rint(rs["c2"]).reshape(len(rs), 8).T[::-1])] if len(rs) else rs


def main():
    out = {}
    for name in sys.argv[1:] or ["stream_1280x720_s1_300", "stream_1920x1080_s0_100", "stream_3840x2160_s4_16"]:
        g = np.load(os.path.join(helpers.GOLDEN, name + ".npz"))
        iw, ih, nframes, tan = int(g["iw"]), int(g["ih"]), int(g["nframes"]), float(g["tan_aov"])
        det = ra.Detector(iw, ih, nslots=1)
        bad = []
        for t in range(nframes):
            det.enqueue(synth.frame(int(g["seed"]), iw, ih, t))
            rects = det.poll(tan)
            segs = det.last_segments()
            ref = g[f"f{t}_rects"]
            seg_ok = helpers.segments_equal(segs, g[f"f{t}_segments"])
            same = len(rects) == len(ref) and helpers.rects_equal(canon(rects), canon(ref))
            if not same or not seg_ok:
                bad.append(t)
                out[f"{name}_f{t}"] = rects
                print(name, "frame", t, "segments equal:", seg_ok, "rects here", len(rects), "reference", len(ref), flush=True)
        print(name, ":", len(bad), "of", nframes, "frames differ:", bad, flush=True)
        det.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "stream_mismatch.npz"), **out)


if __name__ == "__main__":
    main()
