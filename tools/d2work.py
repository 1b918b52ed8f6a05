#!/usr/bin/env python3
"""work-list lengths of the absorption rounds (k_despeckle2_active launches) on frames of the bench stream"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rectdetect_amd as ra
from rectdetect_amd import synth
iw, ih = 1920, 1080
det = ra.Detector(iw, ih, nslots=1)
for t in range(6):
    det.enqueue(synth.frame(synth.SEED0, iw, ih, t))
    det.poll(0.7)
    w = det.plane("d2work", np.int32, 16)
    print("frame", t, "lists:", w[:3].tolist(), "per launch:", w[3:16].tolist())
det.close()
