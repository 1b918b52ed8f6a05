#!/usr/bin/env python3
"""the reference's two-deep call sequence alone (enqueueTask, then pollTask of the frame before), for traces: python tools/two_deep.py [frames = 300]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import rectdetect_amd as ra
from rectdetect_amd import synth
L = ra.lib()
frames = []
for t in range(16):
    a = np.zeros((1080, 1920, 3), np.uint8); L.rd_synth_frame(a.ctypes.data, 1920, 1080, 1920 * 3, synth.SEED0, t, 1); frames.append(a)
bench.pin_to_gpu_cores(L, 0)
ctx = ra.Context(0)
det = ra.RectDetector(ctx, 1920, 1080)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for k in range(8):
    det.execute_once(frames[k % 16], bench.TAN_AOV)
det.enqueue(frames[0])
t0 = time.perf_counter()
for k in range(n):
    det.enqueue(frames[(k + 1) % 16]); det.poll(bench.TAN_AOV)
dt = time.perf_counter() - t0
det.poll(bench.TAN_AOV)
print("two deep: %.1f frames/s" % (n / dt))
det.close(); ctx.close()
