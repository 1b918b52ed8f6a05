#!/usr/bin/env python3
"""3840x2160 stream (BASELINE.json configs[3]): frames/s with 16 frames in flight; prints the polyline phase statistics"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rectdetect_amd as ra
from rectdetect_amd import synth
L = ra.lib()
TAN = float(np.tan(36.0 / 180 * np.pi))
iw, ih, slots = 3840, 2160, int(os.environ.get("SLOTS", "16"))
det = ra.Detector(iw, ih, nslots=slots, nworkers=1)
frames = []
for t in range(16):
    a = np.zeros((ih, iw, 3), np.uint8)
    L.rd_synth_frame(a.ctypes.data, iw, ih, iw * 3, synth.SEED0 + 4, t, 1)
    p = L.rd_device_alloc(a.nbytes); L.rd_upload(p, a.ctypes.data, a.nbytes); frames.append(p)
def run(n):
    infl = 0
    for i in range(n):
        if infl == slots:
            det.poll(TAN); infl -= 1
        det.enqueue(frames[i % 16], ws=iw * 3, on_device=True); infl += 1
    while infl:
        det.poll(TAN); infl -= 1
run(48)
for rep in range(3):
    t0 = time.perf_counter(); run(128); dt = time.perf_counter() - t0
    print("4K: %.1f frames/s, repeated (polyline overflow) %d, env %s" % (128 / dt, det.redone_frames(), {k: v for k, v in os.environ.items() if k.startswith("RD_")}), flush=True)
det.close()
