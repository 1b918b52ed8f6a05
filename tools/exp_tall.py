#!/usr/bin/env python3
"""Experiment (not a test): how would frame-batched launches scale?  The dense stages of B frames stacked into one tall image
(1920 x 1080*B) cost what a launch with B frames per kernel would cost; the polyline stage is left out (RD_DIAG_SKIP=4) because its
single-block kernel does not behave like B blocks would.  Prints pixels/s for B = 1, 4, 8 and several numbers of frames in flight."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
skip = os.environ.get("SKIP", "4")
if skip != "0":
    os.environ["RD_DIAG_SKIP"] = skip
os.environ["RD_DIAG_NO_POST"] = "1"
import rectdetect_amd as ra
from rectdetect_amd import synth
L = ra.lib()
TAN = 0.7
iw = 1920
for B in (1, 4, 8):
    ih = 1080 * B
    a = np.zeros((ih, iw, 3), np.uint8)
    for b in range(B):
        L.rd_synth_frame(a[b * 1080:].ctypes.data, iw, 1080, iw * 3, synth.SEED0 + b, b, 1)
    p = L.rd_device_alloc(a.nbytes); L.rd_upload(p, a.ctypes.data, a.nbytes)
    for slots in (1, 2, 4, 8):
        if B * slots > 32: continue
        det = ra.Detector(iw, ih, nslots=slots, nworkers=1)
        def run(n):
            infl = 0
            for i in range(n):
                if infl == slots:
                    det.poll(TAN); infl -= 1
                det.enqueue(p, ws=iw * 3, on_device=True); infl += 1
            while infl:
                det.poll(TAN); infl -= 1
        run(2 * slots + 2)
        n = max(8, 128 // B)
        t0 = time.perf_counter(); run(n); dt = time.perf_counter() - t0
        print("B=%d (1920x%d) slots=%d: %.1f tall-frames/s = %.1f 1080p-frames/s, budget %s" % (B, ih, slots, n / dt, n * B / dt, det.region_round_budget()), flush=True)
        det.close()
    L.rd_device_free(p)
