#!/usr/bin/env python3
"""Why are host frames 7-8 % slower than resident ones?  Probes on the headline configuration (1920x1080, 64 in flight, groups of 8), all in one small harness:
  PROBE_ORDER=det_first|frames_first   the detector (a 29 GB arena) is created before / after the frames are uploaded
  PROBE_PINNED=0|1                     256 x 6.2 MB of pinned host memory are allocated (before the detector) or not
modes timed: resident; resident while a stream of the script's own uploads 6.2 MB per frame into a buffer nobody reads; host frames from pinned memory (PROBE_PINNED=1).
usage (GPU box): python tools/host_dma_probe.py"""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rectdetect_amd as ra
from rectdetect_amd import synth

IW, IH, F, SLOTS = 1920, 1080, 256, 64
TAN = float(np.tan(36 / 180 * np.pi))
ORDER, PINNED = os.environ.get("PROBE_ORDER", "frames_first"), os.environ.get("PROBE_PINNED", "1") == "1"
L = ra.lib()
hip = ctypes.CDLL("libamdhip64.so")
hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
det = ra.Detector(IW, IH, nslots=SLOTS, nworkers=1) if ORDER == "det_first" else None
frames, dframes, pframes = [], [], []
for t in range(F):
    a = np.zeros((IH, IW, 3), np.uint8)
    L.rd_synth_frame(a.ctypes.data, IW, IH, IW * 3, synth.SEED0, t, 1)
    frames.append(a)
    p = L.rd_device_alloc(a.nbytes); L.rd_upload(p, a.ctypes.data, a.nbytes); dframes.append(p)
    if PINNED:
        q = L.rd_host_alloc(a.nbytes); ctypes.memmove(q, a.ctypes.data, a.nbytes); pframes.append(q)
nbytes = frames[0].nbytes
if det is None:
    det = ra.Detector(IW, IH, nslots=SLOTS, nworkers=1)
st = ctypes.c_void_p()
sink = [L.rd_device_alloc(nbytes) for _ in range(8)]


def run(mode, steps):
    inflight = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        for i in range(F):
            if inflight == SLOTS:
                det.poll(TAN); inflight -= 1
            if mode == "host_pinned":
                det.enqueue(pframes[i], ws=IW * 3, pinned=True)
            else:
                if mode == "resident_plus_copies":
                    assert hip.hipMemcpyAsync(sink[i % 8], pframes[i], nbytes, 1, st) == 0
                det.enqueue(dframes[i], ws=IW * 3, on_device=True)
            inflight += 1
    while inflight:
        det.poll(TAN); inflight -= 1
    det.drain()
    if st: hip.hipStreamSynchronize(st)
    return steps * F / (time.perf_counter() - t0)


run("resident", 4)
modes = ["resident"]
if PINNED:
    modes += ["resident_plus_copies", "host_pinned"]
for rep in range(3):
    for mode in modes:
        if mode == "resident_plus_copies" and not st:
            assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0
        print(ORDER, "pinned_allocs" if PINNED else "no_pinned_allocs", mode, round(run(mode, 8), 1), "frames/s", flush=True)
