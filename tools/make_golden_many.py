#!/usr/bin/env python3
"""A wider set of end-to-end golden results from THE REFERENCE (oracle/_ref): rectangle lists and line-segment lists of many
short synthetic streams (consecutive frames, so the state carried between frames takes part) -> tests/golden/many_rect.npz.
Purpose: the region-labelling stages are order-dependent in the reference and evaluated in a different
schedule here; these frames check that the final outputs nevertheless agree, on far more inputs than the plane-level fixtures."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rectdetect_amd as ra  # noqa: E402
from rectdetect_amd import synth  # noqa: E402
from tests import helpers  # noqa: E402

TAN36 = float(np.tan(36.0 / 180.0 * np.pi))
HARD = [(kind, seed, 640, 480) for kind in ("tiles", "noise", "waves", "bars") for seed in (1, 2, 3)] + [("tiles", 4, 1280, 720), ("waves", 5, 1280, 720)]
STREAMS = [(640, 480, 10 + k, 3) for k in range(16)] + [(1280, 720, 30 + k, 2) for k in range(4)] + [(1920, 1080, 40 + k, 2) for k in range(2)] + [(333, 217, 50 + k, 2) for k in range(4)]


def run_stream(si):
    """one stream in this process (the reference hands out a limited number of kernel ids per process: a fresh process per stream)"""
    iw, ih, seed, nframes = STREAMS[si]
    out = {}
    r = helpers.RefRect(iw, ih)
    for t in range(nframes):
        img = synth.frame(synth.SEED0 + seed, iw, ih, t)
        rects, snaps = r.execute_once(img, TAN36, snapshots=["lslist"])
        n = int(snaps["lslist"][0])
        out["s%d_f%d_rects" % (si, t)] = rects
        out["s%d_f%d_segments" % (si, t)] = snaps["lslist"][: 14 * (n + 1)].view(ra.LS_DTYPE)
        print("stream", si, (iw, ih, seed), "frame", t, "rects", len(rects), "segments", n, flush=True)
    r.close()
    return out


def main():
    import subprocess
    import tempfile
    if len(sys.argv) == 3 and sys.argv[1] == "--stream":
        np.savez(sys.argv[2], **run_stream(int(os.path.basename(sys.argv[2]).split(".")[0])))
        return
    if len(sys.argv) == 3 and sys.argv[1] == "--hard":
        hi = int(os.path.basename(sys.argv[2]).split(".")[0])
        kind, seed, iw, ih = HARD[hi]
        r = helpers.RefRect(iw, ih)
        rects, snaps = r.execute_once(synth.hard_frame(kind, seed, iw, ih), TAN36, snapshots=["lslist"])
        n = int(snaps["lslist"][0])
        print("hard", hi, HARD[hi], "rects", len(rects), "segments", n, flush=True)
        np.savez(sys.argv[2], **{"h%d_rects" % hi: rects, "h%d_segments" % hi: snaps["lslist"][: 14 * (n + 1)].view(ra.LS_DTYPE)})
        return
    out = {"streams": np.array(STREAMS, np.int32), "tan_aov": TAN36}
    with tempfile.TemporaryDirectory() as td:
        for si in range(len(STREAMS)):
            f = os.path.join(td, "%d.npz" % si)
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--stream", f])
            with np.load(f) as z:
                for k in z.files:
                    out[k] = z[k]
    np.savez_compressed(os.path.join(helpers.GOLDEN, "many_rect.npz"), **out)
    hard = {"kinds": np.array([h[0] for h in HARD]), "params": np.array([h[1:] for h in HARD], np.int32), "tan_aov": TAN36}
    with tempfile.TemporaryDirectory() as td:
        for hi in range(len(HARD)):
            f = os.path.join(td, "%d.npz" % hi)
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--hard", f])
            with np.load(f) as z:
                for k in z.files:
                    hard[k] = z[k]
    np.savez_compressed(os.path.join(helpers.GOLDEN, "hard_rect.npz"), **hard)


if __name__ == "__main__":
    main()
