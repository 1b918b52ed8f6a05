#!/usr/bin/env python3
"""where a single frame's latency goes (one frame in flight, host buffers): enqueue (copy + upload + launches), poll (wait + post-process), device interval"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rectdetect_amd as ra
from rectdetect_amd import synth
L = ra.lib()
TAN = float(np.tan(36.0 / 180 * np.pi))
frames = []
for t in range(16):
    a = np.zeros((1080, 1920, 3), np.uint8); L.rd_synth_frame(a.ctypes.data, 1920, 1080, 1920 * 3, synth.SEED0, t, 1); frames.append(a)
for label, post in (("host post-process", 0), ("device post-process", 1))[:1 if os.environ.get("LAT_HOST_POST_ONLY") else 2]:
    os.environ["RD_DEVICE_POST"] = str(post)
    det = ra.Detector(1920, 1080, nslots=1, nworkers=0, aperture=TAN)
    for k in range(8):
        det.enqueue(frames[k]); det.poll(TAN)
    te = tp = 0.0; d0 = det.device_time()
    n = 200
    for k in range(n):
        t0 = time.perf_counter(); det.enqueue(frames[k % 16]); t1 = time.perf_counter(); det.poll(TAN); t2 = time.perf_counter()
        te += t1 - t0; tp += t2 - t1
    d1 = det.device_time()
    print("%s: enqueue %.3f ms, poll %.3f ms, total %.3f ms; device interval %.3f ms" % (label, 1e3 * te / n, 1e3 * tp / n, 1e3 * (te + tp) / n, (d1[0] - d0[0]) / max(1, d1[1] - d0[1]) / 1e3))
    det.close()
