#!/usr/bin/env python3
"""The two reference-API rows of the bench line alone (bench.py: reference_api_config): oclrect_executeOnce per frame and enqueueTask / pollTask two deep, host buffers.
python tools/api_rate.py [seconds per leg = 1.5]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import rectdetect_amd as ra  # noqa: E402
from rectdetect_amd import synth  # noqa: E402

L = ra.lib()
frames = []
for t in range(32):
    a = np.zeros((bench.IH, bench.IW, 3), np.uint8)
    L.rd_synth_frame(a.ctypes.data, bench.IW, bench.IH, bench.IW * 3, synth.SEED0, t, 1)
    frames.append(a)
bench.pin_to_gpu_cores(L, 0)
two, once = bench.reference_api_config(ra, frames, 0, float(sys.argv[1]) if len(sys.argv) > 1 else 1.5)
print("two deep %.1f frames/s | executeOnce %.1f frames/s, median %.3f ms, p90 %.3f ms" % (two["value"], once["value"], once["latency_ms_median"], once["latency_ms_p90"]))
