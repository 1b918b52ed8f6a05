#!/usr/bin/env python3
"""Generate tests/golden/*.npz from THE REFERENCE ITSELF (oracle/_ref: its kernels and host C on the serial OpenCL shim).
Only runs where /root/reference exists; the fixtures it writes are data (inputs are re-created from seeds by
rectdetect_amd/synth.py): rectangles, line segments, convergence flags and CRC32s of intermediate planes."""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rectdetect_amd as ra
from rectdetect_amd import synth
from tests import helpers

TAN36 = float(np.tan(36.0 / 180.0 * np.pi))
RECT_CASES = [  # name, iw, ih, seed offset, frames (consecutive frames of one stream: exercises the state carried between frames)
    ("rect_640x480_s0", 640, 480, 0, 2),
    ("rect_640x480_s5", 640, 480, 5, 1),
    ("rect_333x217_s2", 333, 217, 2, 1),
    ("rect_1280x720_s1", 1280, 720, 1, 2),
    ("rect_1920x1080_s0", 1920, 1080, 0, 1),
]
POLY_CASES = [("poly_640x480_s0", 640, 480, 0, 500, 1.0, 20), ("poly_333x217_s2", 333, 217, 2, 500, 1.0, 20), ("poly_1280x720_s1_vid", 1280, 720, 1, 2000, 1.0, 10)]
PLANES = ["plab0", "Lblur", "plab1", "vxy", "strength", "nms", "mask0", "tidy", "str_sum", "edge500", "smooth", "quant", "strong", "label1", "junction",
          "mergemask", "rsize", "region", "boundary_src", "boundary", "lsid", "table"]


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def main():
    os.makedirs(helpers.GOLDEN, exist_ok=True)
    for name, iw, ih, seed, nframes in RECT_CASES:
        N = iw * ih
        r = helpers.RefRect(iw, ih)
        out = {"iw": iw, "ih": ih, "seed": synth.SEED0 + seed, "nframes": nframes, "tan_aov": TAN36}
        for t in range(nframes):
            img = synth.frame(synth.SEED0 + seed, iw, ih, t)
            rects, snaps = r.execute_once(img, TAN36, snapshots=PLANES + ["lslist", "flags_label1", "flags_boundary", "flags_chain", "flags_sub"])
            n = int(snaps["lslist"][0])
            segs = snaps["lslist"][: 14 * (n + 1)].view(ra.LS_DTYPE)
            out[f"f{t}_input_crc"] = crc(img)
            out[f"f{t}_rects"] = rects
            out[f"f{t}_segments"] = segs
            out[f"f{t}_launches"] = r.launches
            out[f"f{t}_flags"] = np.stack([snaps[k][:13].view("i4") for k in ("flags_label1", "flags_boundary", "flags_chain", "flags_sub")])
            sizes = {"vxy": 2 * N, "table": (N * 4 // 5) * 5}
            out[f"f{t}_plane_crc"] = np.array([crc(snaps[p][: sizes.get(p, N)]) for p in PLANES], np.uint32)
            print(name, t, "rects", len(rects), "segments", n, "launches", r.launches)
        out["planes"] = np.array(PLANES)
        np.savez_compressed(os.path.join(helpers.GOLDEN, name + ".npz"), **out)
        r.close()
    R = helpers.ref()
    for name, iw, ih, seed, sthr, minerr, sizethr in POLY_CASES:
        N = iw * ih
        img = synth.frame(synth.SEED0 + seed, iw, ih, 0)
        ls = np.zeros(N * 4, np.int32)
        ids = np.zeros(N, np.int32)
        R.rdcl_trace_reset()
        n = R.rdref_poly_run(img.ctypes.data, iw, ih, iw * 3, sthr, minerr, sizethr, helpers.P(ls), helpers.P(ids), None)
        segs = ls[: 14 * (n + 1)].view(ra.LS_DTYPE)
        np.savez_compressed(os.path.join(helpers.GOLDEN, name + ".npz"), iw=iw, ih=ih, seed=synth.SEED0 + seed, strength_thre=sthr, minerror=minerr,
                            size_thre=sizethr, input_crc=crc(img), segments=segs, ids_crc=crc(ids), launches=R.rdcl_trace_count())
        print(name, "segments", n, "launches", R.rdcl_trace_count())


if __name__ == "__main__":
    main()
