#!/usr/bin/env python3
"""throughput at several frame sizes (frames resident in HBM, 16 in flight): separates per-pixel work from per-launch cost"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rectdetect_amd as ra
from rectdetect_amd import synth
L = ra.lib()
TAN = float(np.tan(36.0 / 180 * np.pi))
for iw, ih in [(640, 480), (1280, 720), (1920, 1080), (3840, 2160)]:
    slots = int(os.environ.get("SLOTS", "16"))
    if iw * ih > 1920 * 1088: slots = min(slots, 16)
    det = ra.Detector(iw, ih, nslots=slots, nworkers=1)
    frames = []
    for t in range(16):
        a = np.zeros((ih, iw, 3), np.uint8)
        L.rd_synth_frame(a.ctypes.data, iw, ih, iw * 3, synth.SEED0, t, 1)
        p = L.rd_device_alloc(a.nbytes); L.rd_upload(p, a.ctypes.data, a.nbytes); frames.append(p)
    def run(n):
        infl = 0
        for i in range(n):
            if infl == slots:
                det.poll(TAN); infl -= 1
            det.enqueue(frames[i % 16], ws=iw * 3, on_device=True); infl += 1
        while infl:
            det.poll(TAN); infl -= 1
    run(32)
    n = 256 if iw < 3000 else 64
    t0 = time.perf_counter(); run(n); dt = time.perf_counter() - t0
    print("%dx%d: %.1f fps, %.2f Gpixel/s, redone %d" % (iw, ih, n / dt, n / dt * iw * ih / 1e9, det.redone_frames()), flush=True)
    det.close()
    for p in frames: L.rd_device_free(p)
