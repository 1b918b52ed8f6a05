#!/usr/bin/env python3
"""Stage-by-stage comparison of the HIP path with the oracle on synthetic frames (run on the GPU box)."""
import sys
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rectdetect_amd as ra
from rectdetect_amd import synth
from tests import helpers

PAIRS = [("plab0", "plab0", 1), ("lblur", "Lblur", 1), ("plab1", "plab1", 1), ("vxy", "vxy", 2), ("strength", "strength", 1), ("nms", "nms", 1),
         ("mask0", "mask0", 1), ("tidy", "tidy", 1), ("strsum", "str_sum", 1), ("edge500", "edge500", 1), ("smooth", "smooth", 1), ("quant", "quant", 1),
         ("strong", "strong", 1), ("label1", "label1", 1), ("junction", "junction", 1), ("mergemask", "mergemask", 1), ("region", "region", 1),
         ("rsize", "rsize", 1), ("boundarysrc", "boundary_src", 1), ("boundary", "boundary", 1), ("lsid", "lsid", 1)]


def main():
    cases = [(640, 480, 0, 2), (1280, 720, 1, 1), (1920, 1080, 0, 1)]
    if len(sys.argv) > 1:
        cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
    print(ra.lib().rd_version().decode(), "devices", ra.lib().rd_device_count())
    for iw, ih, seed, nframes in cases:
        N = iw * ih
        det = ra.Detector(iw, ih, nslots=1)
        orc = helpers.OracleRect(iw, ih)
        for t in range(nframes):
            img = synth.frame(synth.SEED0 + seed, iw, ih, t)
            t0 = time.time()
            det.enqueue(img)
            rects = det.poll(np.tan(np.pi / 5))
            t1 = time.time()
            orc.frame(img)
            print(f"== {iw}x{ih} seed {seed} t {t}: hip {1e3*(t1-t0):.1f} ms, oracle {time.time()-t1:.2f} s, rects {len(rects)}")
            for g, o, k in PAIRS:
                a = det.plane(g, np.uint32, N * k)
                b = orc.plane(o).view(np.uint32)[: N * k]
                d = a != b
                msg = f"   {g:12s} mismatches {int(d.sum()):8d}"
                if d.any():
                    i = np.nonzero(d)[0][:4]
                    msg += "  first " + str([(int(j % (iw * k)) // k, int(j // (iw * k)), hex(int(a[j])), hex(int(b[j]))) for j in i])
                print(msg)
            segs, osegs = det.last_segments(), orc.segments()
            print("   segments n", int(segs.view('i4')[0]), "oracle n", int(osegs.view('i4')[0]), "identical", helpers.segments_equal(segs, osegs))
            tb = det.plane("table", np.int32, (N * 4 // 5) * 5)
            otb = orc.plane("table")[: (N * 4 // 5) * 5]
            print("   table mismatches", int((tb != otb).sum()))
            orects = ra.postprocess_planes(osegs, orc.plane("boundary"), orc.plane("table"), iw, ih, np.tan(np.pi / 5))
            print("   rects hip", len(rects), "oracle", len(orects), "identical", helpers.rects_equal(rects, orects))
        det.close()
        orc.close()


if __name__ == "__main__":
    main()
