#!/usr/bin/env python3
"""one detector, 3840x2160 frames resident in HBM, a few frames: for kernel traces at a size where every dense kernel fills the chip"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rectdetect_amd as ra
from rectdetect_amd import synth
L = ra.lib()
iw, ih = 3840, 2160
det = ra.Detector(iw, ih, nslots=1, nworkers=0)
frames = []
for t in range(4):
    a = np.zeros((ih, iw, 3), np.uint8)
    L.rd_synth_frame(a.ctypes.data, iw, ih, iw * 3, synth.SEED0, t, 1)
    p = L.rd_device_alloc(a.nbytes); L.rd_upload(p, a.ctypes.data, a.nbytes); frames.append(p)
TAN = float(np.tan(36.0 / 180 * np.pi))
t0 = time.perf_counter()
for i in range(8):
    det.enqueue(frames[i % 4], ws=iw * 3, on_device=True); det.poll(TAN)
print("4K single stream: %.1f fps" % (8 / (time.perf_counter() - t0)))
det.close()
