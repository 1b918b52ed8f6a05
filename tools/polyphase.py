#!/usr/bin/env python3
"""Print the phase durations of the single-launch polyline kernel (100 MHz stamps in the polyline counters) for a few frames."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rectdetect_amd as rd
from rectdetect_amd import synth

iw, ih = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
det = rd.Detector(iw, ih, nslots=1)
for f in range(4):
    det.enqueue(synth.frame(synth.SEED0, iw, ih, f))
    det.poll(1.0)
    c = det.plane("polyctr", count=64)
    t = c[39:46].astype(np.int64)
    nr = int(c[38])
    rs = np.append(c[46:46 + nr].astype(np.int64), t[3])
    print(f, "chain pixels", int(c[0]), "chains", int(c[1]), "live", int(c[24]), "phases us:", (np.diff(t) / 100.0).round(1).tolist(), "total", (t[-1] - t[0]) / 100.0,
          "| rounds", nr, "us each:", ((np.diff(rs) & 0xffffffff) / 100.0).round(1).tolist())
det.close()
