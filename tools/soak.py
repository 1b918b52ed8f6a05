#!/usr/bin/env python3
"""Soak / determinism check: the same synthetic stream twice through an 16-slot detector (frames resident in HBM) and once through a
1-slot detector; the rectangle lists of all three runs must be identical frame by frame."""
import sys, os, hashlib, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rectdetect_amd as ra
from rectdetect_amd import synth

nframes = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iw, ih = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
L = ra.lib()
TAN = float(np.tan(36.0 / 180 * np.pi))
frames = []
for t in range(64):
    a = np.zeros((ih, iw, 3), np.uint8)
    L.rd_synth_frame(a.ctypes.data, iw, ih, iw * 3, synth.SEED0 + 5, t, 1)
    p = L.rd_device_alloc(a.nbytes); L.rd_upload(p, a.ctypes.data, a.nbytes); frames.append(p)


def run(slots, workers=1):
    det = ra.Detector(iw, ih, nslots=slots, nworkers=workers)
    out, infl = [], 0
    t0 = time.perf_counter()
    for i in range(nframes):
        if infl == slots:
            out.append(hashlib.md5(np.asarray(det.poll(TAN)).tobytes()).hexdigest()); infl -= 1
        det.enqueue(frames[i % 64], ws=iw * 3, on_device=True); infl += 1
    while infl:
        out.append(hashlib.md5(np.asarray(det.poll(TAN)).tobytes()).hexdigest()); infl -= 1
    dt = time.perf_counter() - t0
    rb = det.region_round_budget()
    det.close()
    return out, nframes / dt, rb


SL = int(os.environ.get("SLOTS", "16"))      # (16: sparse stages batched, launch budgets that change in mid-run)
a, fa, ra_ = run(SL)
b, fb, rb_ = run(SL)
c, fc, rc_ = run(1)
print("frames", nframes, "fps", round(fa), round(fb), round(fc), "round budget / repeats", ra_, rb_, rc_)
d, fd, rd_ = run(2, 0)      # the reference's call shape: two frames in flight, the caller's thread post-processes - with the helper threads of rd_post.c
print(SL, "slots twice identical:", a == b, "|", SL, "slots vs 1 slot identical:", a == c, "| vs 2 slots without workers (%d frames/s, %d helper threads) identical:" % (round(fd), L.rd_post_helpers()), a == d)
sys.exit(0 if (a == b and a == c and a == d) else 1)
