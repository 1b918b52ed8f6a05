#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (a probe, not a test).  Does the reference's labelling - ten fixed passes of in-place min-propagation (oclimgutil.c:227-246) - settle on a REAL,
parallel OpenCL device?  Runs oclimgutil_label8x_int_int (the reference's unchanged host C, oracle/_ref/librdref_ocl.so) on random 0/1 masks with long thin components and on a
synthetic frame's edge mask, and compares with the settled labelling (8-connected components of equal value, label = smallest pixel index; scipy).  Pixels whose label is not the
settled one = the labelling had not converged after its ten passes under this device's work-item order.
usage (GPU box): python tools/ref_label_on_opencl.py -> gpurun_out/ref_label_opencl.json"""
import ctypes
import json
import os
import sys

import numpy as np
from scipy import ndimage

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden_ops as mg  # noqa: E402
from rectdetect_amd import synth  # noqa: E402

vp, ci = ctypes.c_void_p, ctypes.c_int


def settled(mask, bgc):
    ih, iw = mask.shape
    out = np.full(mask.shape, -1, np.int64)
    idx = np.arange(ih * iw).reshape(ih, iw)
    for v in np.unique(mask):
        if v == bgc:
            continue
        lab, n = ndimage.label(mask == v, structure=np.ones((3, 3)))
        mins = ndimage.minimum(idx, lab, index=np.arange(1, n + 1))
        sel = lab > 0
        out[sel] = np.asarray(mins, np.int64)[lab[sel] - 1]
    return out


def main():
    o = mg.Ops(ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "librdref_ocl.so")))
    rng = np.random.default_rng(5)
    cases = {}
    # (a) sparse random mask, (b) long diagonal / serpentine lines (chains a raster sweep settles in one pass and a parallel one does not), (c) a frame-like edge mask: the borders of the synthetic quads
    iw, ih = 640, 480
    m = (rng.random((ih, iw)) < 0.35).astype(np.int32)
    cases["random 35 % of 640x480"] = (m, -1)
    s = np.zeros((ih, iw), np.int32)
    for k in range(0, ih - 8, 8):
        s[k, 4:iw - 4] = 1
        s[k:k + 8, (iw - 5) if (k // 8) % 2 == 0 else 4] = 1
    cases["one serpentine line through 640x480"] = (s, -1)
    iw2, ih2 = 1920, 1080
    f = synth.frame(synth.SEED0, iw2, ih2, 0).astype(np.int32).sum(2)
    e = ((np.abs(np.diff(f, axis=0, prepend=f[:1])) + np.abs(np.diff(f, axis=1, prepend=f[:, :1]))) > 40).astype(np.int32)
    cases["quad borders of the 1920x1080 bench frame (background labelled too)"] = (e, -1)
    cases["the same, background = 0 left out"] = (e, 0)
    rep = {}
    for name, (mask, bgc) in cases.items():
        h, w = mask.shape
        N = w * h
        mi, mo, mt = o.buf(mask), o.buf(N * 4), o.buf(N * 4)
        o.call("label8x_int_int", [vp, vp, vp, ci, ci, ci], mo, mi, mt, bgc, w, h)
        got = o.read(mo, np.int32, N).reshape(h, w).astype(np.int64)
        want = settled(mask, bgc)
        bad = int((got != want).sum())
        rep[name] = {"pixels": N, "labelled": int((want >= 0).sum()), "pixels_not_at_the_settled_label": bad}
        print(name, rep[name], flush=True)
        for b in (mi, mo, mt):
            o.L.clReleaseMemObject(b)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rep, open(os.path.join(ROOT, "gpurun_out", "ref_label_opencl.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
