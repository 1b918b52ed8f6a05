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
import ctypes
frames, hframes, pframes = [], [], []
for t in range(64):
    a = np.zeros((ih, iw, 3), np.uint8)
    L.rd_synth_frame(a.ctypes.data, iw, ih, iw * 3, synth.SEED0 + 5, t, 1)
    p = L.rd_device_alloc(a.nbytes); L.rd_upload(p, a.ctypes.data, a.nbytes); frames.append(p)
    hframes.append(a)
    q = L.rd_host_alloc(a.nbytes); ctypes.memmove(q, a.ctypes.data, a.nbytes); pframes.append(q)


def run(slots, workers=1, where="resident"):
    """where: resident (frames in HBM), pinned (RD_FRAME_HOST_PINNED: read in place by the copy engine), pageable (copied by the caller's thread first)"""
    det = ra.Detector(iw, ih, nslots=slots, nworkers=workers)
    out, infl = [], 0
    t0 = time.perf_counter()
    for i in range(nframes):
        if infl == slots:
            out.append(hashlib.md5(np.asarray(det.poll(TAN)).tobytes()).hexdigest()); infl -= 1
        if where == "resident": det.enqueue(frames[i % 64], ws=iw * 3, on_device=True)
        elif where == "pinned": det.enqueue(pframes[i % 64], ws=iw * 3, pinned=True)
        else: det.enqueue(hframes[i % 64])
        infl += 1
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
e, fe, _ = run(SL, 1, "pinned")
f, ff, _ = run(SL, 1, "pageable")
print("host frames handed over,", SL, "slots: pinned %d frames/s identical: %s | pageable %d frames/s identical: %s" % (round(fe), a == e, round(ff), a == f))
if not (a == e and a == f): sys.exit(1)
print(SL, "slots twice identical:", a == b, "|", SL, "slots vs 1 slot identical:", a == c, "| vs 2 slots without workers (%d frames/s, %d helper threads) identical:" % (round(fd), L.rd_post_helpers()), a == d)
sys.exit(0 if (a == b and a == c and a == d) else 1)
