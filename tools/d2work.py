#!/usr/bin/env python3
"""Print the work-list length of every despeckle2 launch (diagnostic) for a few frames of the synthetic stream."""
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
    print(f, det.plane("d2work", count=16).tolist())
det.close()
