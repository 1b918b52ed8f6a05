#!/usr/bin/env python3
"""Where does the caller's thread spend a frame in the reference's two-deep loop (oclrect_enqueueTask / oclrect_pollTask, vidrect.cpp:159-205)?  Wall time inside enqueue, inside
poll, of that the host post-process (counter 13), and the frame period.  usage (GPU box): python tools/two_deep_host.py [pinned]"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rectdetect_amd as ra
from rectdetect_amd import synth
IW, IH = 1920, 1080
TAN = float(np.tan(36 / 180 * np.pi))
L = ra.lib()
ctx = ra.Context(0)
det = ra.RectDetector(ctx, IW, IH)
frames = []
for t in range(16):
    a = np.zeros((IH, IW, 3), np.uint8)
    L.rd_synth_frame(a.ctypes.data, IW, IH, IW * 3, synth.SEED0, t, 1)
    frames.append(ctx.pinned_copy(a) if len(sys.argv) > 1 else a)
d = ctypes.cast(det.h + 8, ctypes.POINTER(ctypes.c_void_p))[0]
for k in range(8):
    det.execute_once(frames[k], TAN)
det.enqueue(frames[0])
n = 1500
te = tp = 0.0
c0 = [L.rd_detector_counter(d, k) for k in (3, 13, 1, 2)]
t0 = time.perf_counter()
for k in range(n):
    a = time.perf_counter(); det.enqueue(frames[(k + 1) % 16]); b = time.perf_counter(); det.poll(TAN); c = time.perf_counter()
    te += b - a; tp += c - b
dt = time.perf_counter() - t0
c1 = [L.rd_detector_counter(d, k) for k in (3, 13, 1, 2)]
det.poll(TAN)
print("frames/s %.1f  period %.1f us | in enqueue %.1f us (library's own count %.1f) | in poll %.1f us, of that host post-process %.1f | device interval per frame %.1f us" % (
    n / dt, 1e6 * dt / n, 1e6 * te / n, (c1[0] - c0[0]) / n, 1e6 * tp / n, (c1[1] - c0[1]) / n, (c1[2] - c0[2]) / max(1, c1[3] - c0[3])))
