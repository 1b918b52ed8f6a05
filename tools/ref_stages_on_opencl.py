#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Stage by stage: the planes the reference produces on the box's REAL OpenCL device (its unchanged host C and .cl sources, oracle/_ref/librdref_ocl.so; the
observer refshim/rdcl_observe.c reads a buffer back right after the launch that completes it - the same launches the serial stand-in's snapshots name, tests/helpers.py:
REF_SNAPSHOTS) against the oracle's planes (oracle/librd_oracle.so, reference mode = bit-identical to the reference on the stand-in: tests/test_cpu_oracle.py), on single frames.
Per plane: elements that differ in any bit.  With AMD_OCL_BUILD_OPTIONS_APPEND carrying the goldens' arithmetic contract and the pinned builtins (tools/gpu_probe_ocl5.sh) what
differs is what the device's work-item order makes of the reference's in-place kernels.
usage (GPU box): python tools/ref_stages_on_opencl.py [tag] -> gpurun_out/ref_stages_opencl_<tag>.json"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rectdetect_amd import RECT_DTYPE, synth  # noqa: E402
from tests import helpers  # noqa: E402

TAN36 = float(np.tan(36.0 / 180.0 * np.pi))
ORDER = ["plab0", "Lblur", "plab1", "vxy", "strength", "nms", "mask0", "tidy", "str_sum", "edge500", "smooth", "quant", "strong", "label1", "junction", "mergemask", "rsize", "region",
         "boundary_src", "boundary", "lsid", "lslist", "table"]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "default"
    R = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "librdref_ocl.so"))
    R.rdref_rect_open.restype = ctypes.c_void_p
    R.rdref_rect_open.argtypes = [ctypes.c_int, ctypes.c_int]
    R.rdref_rect_close.argtypes = [ctypes.c_void_p]
    R.rdref_rect_execute_once.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_int]
    R.rdcl_snapshot_request.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    R.rdcl_snapshot_fetch.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_int)]
    rep = {"build_options_appended": os.environ.get("AMD_OCL_BUILD_OPTIONS_APPEND", ""), "frames": {}}
    for seed, iw, ih, t in [(0, 640, 480, 0), (5, 640, 480, 0), (1, 1280, 720, 0), (0, 1920, 1080, 0), (0, 1920, 1080, 1)]:
        N = iw * ih
        img = np.ascontiguousarray(synth.frame(synth.SEED0 + seed, iw, ih, t)).copy()
        h = R.rdref_rect_open(iw, ih)
        R.rdcl_trace_reset()
        R.rdcl_snapshot_clear()
        hs = {k: R.rdcl_snapshot_request(helpers.REF_SNAPSHOTS[k][0].encode(), helpers.REF_SNAPSHOTS[k][1], helpers.REF_SNAPSHOTS[k][2]) for k in ORDER}
        out = np.zeros(1024, RECT_DTYPE)
        n = R.rdref_rect_execute_once(h, img.ctypes.data, img.strides[0], TAN36, out.ctypes.data, 1024)
        launches = R.rdcl_trace_count()
        snaps = {}
        for k, hd in hs.items():
            p, sz, o = ctypes.c_void_p(), ctypes.c_size_t(), ctypes.c_int()
            if R.rdcl_snapshot_fetch(hd, ctypes.byref(p), ctypes.byref(sz), ctypes.byref(o)) == 0:
                snaps[k] = np.frombuffer((ctypes.c_char * sz.value).from_address(p.value), dtype="u4").copy()
        R.rdref_rect_close(h)
        orc = helpers.OracleRect(iw, ih)
        orc.frame(img)
        sizes = {"vxy": 2 * N, "table": (N * 4 // 5) * 5}
        rows = {}
        for k in ORDER:
            if k not in snaps:
                rows[k] = "no snapshot"
                continue
            want = orc.plane(k).view(np.uint32)
            m = sizes.get(k, N)
            if k == "lslist":
                m = 14 * (int(want[0]) + 1) if int(want[0]) == int(snaps[k][0]) else 1
            rows[k] = {"elements": int(m), "differing": int((want[:m] != snaps[k][:m]).sum())}
        orc.close()
        name = "seed %d %dx%d t %d" % (seed, iw, ih, t)
        rep["frames"][name] = {"launches": launches, "rectangles": max(0, n - 1), "planes": rows}
        same = [k for k in ORDER if isinstance(rows[k], dict) and rows[k]["differing"] == 0]
        print(name, "launches", launches, "| planes identical in every bit:", len(same), "of", len(ORDER), "| differing:", {k: (rows[k]["differing"] if isinstance(rows[k], dict) else rows[k]) for k in ORDER if k not in same}, flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rep, open(os.path.join(ROOT, "gpurun_out", "ref_stages_opencl_%s.json" % tag), "w"), indent=1)


if __name__ == "__main__":
    main()
