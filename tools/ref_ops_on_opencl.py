#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Operator by operator: the reference's oclimgutil_* operators (its unchanged host C, oracle/_ref/librdref_ocl.so) on the box's REAL OpenCL device against the
operator goldens (tests/golden/ops_*.npz, ops_iir_*.npz: the reference on our serial stand-in, -ffp-contract=off), on the goldens' own inputs.  Per operator: elements that differ in
any bit, largest absolute difference.  With AMD_OCL_BUILD_OPTIONS_APPEND="-Wf,-ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt" the vendor's compiler works under the goldens'
arithmetic contract (tools/gpu_probe_ocl2.sh runs both).  usage (GPU box): python tools/ref_ops_on_opencl.py [tag] -> gpurun_out/ref_ops_opencl_<tag>.json"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden_ops as mg  # noqa: E402   (the operator sequences the goldens were made with)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def diff(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return {"shape": [list(a.shape), list(b.shape)]}
    ne = a.view(np.uint8).reshape(a.size, -1) != b.view(np.uint8).reshape(b.size, -1) if a.dtype.itemsize > 1 else (a != b).reshape(a.size, 1)
    nd = int(ne.any(1).sum())
    out = {"elements": int(a.size), "differing": nd}
    if nd:
        fa, fb = a.astype(np.float64), b.astype(np.float64)
        ok = np.isfinite(fa) & np.isfinite(fb)
        out["max_abs_difference"] = float(np.abs(fa - fb)[ok].max()) if ok.any() else None
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "default"
    o = mg.Ops(ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "librdref_ocl.so")))
    rep = {"build_options_appended": os.environ.get("AMD_OCL_BUILD_OPTIONS_APPEND", ""), "fixtures": {}}
    for name, runner in (("ops_97x61", mg.run_all), ("ops_160x131", mg.run_all), ("ops_iir_97x61", mg.run_iir_radii), ("ops_iir_160x131", mg.run_iir_radii)):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        res = runner(o, int(z["iw"]), int(z["ih"]), int(z["seed"]))
        rows = {}
        for k in sorted(res):
            if k.startswith("in_"):
                assert np.array_equal(res[k], z[k]), "inputs differ"      # (the same seeded inputs as the goldens)
                continue
            rows[k] = diff(res[k], z[k])
        rep["fixtures"][name] = rows
        same = [k for k, v in rows.items() if v.get("differing") == 0]
        print(name, "bit-identical:", len(same), "of", len(rows), "| differing:", {k: (v.get("differing"), v.get("max_abs_difference")) for k, v in rows.items() if v.get("differing") != 0}, flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "ref_ops_opencl_%s.json" % tag), "w") as f:
        json.dump(rep, f, indent=1)


if __name__ == "__main__":
    main()
