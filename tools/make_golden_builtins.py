#!/usr/bin/env python3
"""How soft is the oracle?  The reference runs here on OUR OpenCL stand-in, and the stand-in's builtins make choices the OpenCL standard leaves
to the device (rsqrt, hypot, distance, FMA contraction).  This tool runs THE REFERENCE (oracle/_ref) on fixture frames again
with the other legal choices (oracle/Makefile: ref_variants; oracle/refshim/rdcl_builtins.c: RDCL_VARIANT) and records, per frame and variant,
what moves against the baseline build: pixels of the intermediate planes, segment records, rectangles.

-> tests/golden/builtin_sensitivity.npz: per frame f, `f_union` = the distinct rectangles the reference produced under any variant (variant 0 =
   the baseline of all other fixtures), `f_member[variant, i]` = whether rectangle i of the union is in that variant's list, `f_planes[variant, p]` =
   pixels of plane p (names in `planes`) that differ from the baseline, `f_segs[variant]` = (segment records, records not bit-identical to the
   baseline's, largest coordinate difference among records that pair up by position).
-> tests/golden/builtin_sensitivity.json: the same numbers, readable.
Only runs where /root/reference exists (oracle/_ref); work-item order = serial raster order throughout."""
import json
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rectdetect_amd as ra  # noqa: E402
from rectdetect_amd import synth  # noqa: E402
from tests import helpers  # noqa: E402

TAN36 = float(np.tan(36.0 / 180.0 * np.pi))
VARIANTS = ["baseline", "rsqrt1", "hypotf", "distd", "fma", "all"]      # oracle/Makefile: VARIANTS ("all" = the three builtins + contraction)
PLANES = ["Lblur", "vxy", "strength", "nms", "tidy", "strong", "smooth", "quant", "region", "boundary"]
# (seed offset, iw, ih, t): the stills of the rect fixtures and the first frames of the bench stream
FRAMES = [(0, 640, 480, 0), (5, 640, 480, 0), (1, 1280, 720, 0), (0, 1920, 1080, 0), (0, 1920, 1080, 1), (0, 1920, 1080, 2), (7, 1920, 1080, 0)]


def rect_key(r):
    return r["c2"].tobytes() + r["c3"].tobytes() + r["value"].tobytes() + r["status"].tobytes()


def within_tolerance(a, b, tol=1e-4):
    """the north_star's parity bar between two rectangle lists (as sets): same count and status, integer pixel coordinates identical, floats within tol"""
    if len(a) != len(b):
        return False
    if len(a) == 0:
        return True
    canon = lambda rs: rs[np.lexsort(np.rint(rs["c2"]).reshape(len(rs), 8).T[::-1])]
    a, b = canon(a), canon(b)
    return bool(np.array_equal(a["status"], b["status"]) and np.array_equal(np.rint(a["c2"]), np.rint(b["c2"])) and np.abs(a["c2"] - b["c2"]).max() <= tol and
                np.abs(a["c3"] - b["c3"]).max() <= tol and np.abs(a["value"] - b["value"]).max() <= tol)


def run_one(fi, vi, path):
    seed, iw, ih, t = FRAMES[fi]
    r = helpers.RefRect(iw, ih)
    rects, snaps = r.execute_once(synth.frame(synth.SEED0 + seed, iw, ih, t), TAN36, snapshots=PLANES + ["lslist"])
    n = int(snaps["lslist"][0])
    np.savez(path, rects=rects, segs=snaps["lslist"][: 14 * (n + 1)], **{p: snaps[p] for p in PLANES})
    r.close()


def main():
    if len(sys.argv) == 5 and sys.argv[1] == "--one":
        run_one(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
        return
    refdir = os.path.join(ROOT, "oracle", "_ref")
    out = {"variants": np.array(VARIANTS), "planes": np.array(PLANES), "frames": np.array(FRAMES, np.int32)}
    summary = {"variants": VARIANTS, "planes": PLANES, "frames": []}
    with tempfile.TemporaryDirectory() as td:
        def job(a):
            fi, vi = a
            f = os.path.join(td, "%d_%d.npz" % (fi, vi))
            env = dict(os.environ)
            if vi > 0:
                env["RDCL_KERNEL_DIR"] = os.path.join(refdir, "var_" + VARIANTS[vi])
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--one", str(fi), str(vi), f], stdout=subprocess.DEVNULL, env=env)
            with np.load(f) as z:
                return {k: z[k] for k in z.files}
        with ThreadPoolExecutor(6) as ex:
            res = list(ex.map(job, [(fi, vi) for fi in range(len(FRAMES)) for vi in range(len(VARIANTS))]))
    for fi, fr in enumerate(FRAMES):
        runs = res[fi * len(VARIANTS):(fi + 1) * len(VARIANTS)]
        base = runs[0]
        union, index = [], {}
        for r in runs:
            for q in r["rects"]:
                if rect_key(q) not in index:
                    index[rect_key(q)] = len(union)
                    union.append(q)
        member = np.zeros((len(VARIANTS), len(union)), np.uint8)
        planes = np.zeros((len(VARIANTS), len(PLANES)), np.int64)
        segs = np.zeros((len(VARIANTS), 3), np.float64)
        bs = base["segs"].view(ra.LS_DTYPE)
        for vi, r in enumerate(runs):
            for q in r["rects"]:
                member[vi, index[rect_key(q)]] = 1
            for pi, p in enumerate(PLANES):
                a, b = r[p], base[p]
                planes[vi, pi] = int((a != b).sum()) if a.shape == b.shape else -1
            s = r["segs"].view(ra.LS_DTYPE)
            n, nb = int(s.view("i4")[0]), int(bs.view("i4")[0])
            m = min(len(s), len(bs))
            valid = (s[1:m]["polyid"] != 0) & (bs[1:m]["polyid"] != 0)
            ne = 0 if helpers.segments_equal(s, bs) else int(sum(s[1:m][valid][k].tobytes() != bs[1:m][valid][k].tobytes() for k in range(int(valid.sum())))) + abs(n - nb)
            dmax = max((float(np.abs(s[1:m][valid][c] - bs[1:m][valid][c]).max(initial=0)) for c in ("x0", "y0", "x1", "y1")), default=0.0) if n == nb else -1.0
            segs[vi] = (n, ne, dmax)
        out["f%d_union" % fi] = np.array(union, dtype=ra.RECT_DTYPE) if union else np.zeros(0, ra.RECT_DTYPE)
        out["f%d_member" % fi] = member
        out["f%d_planes" % fi] = planes
        out["f%d_segs" % fi] = segs
        out["f%d_within" % fi] = np.array([within_tolerance(r["rects"], runs[0]["rects"]) for r in runs], np.uint8)
        row = {"frame": {"seed": int(fr[0]), "iw": int(fr[1]), "ih": int(fr[2]), "t": int(fr[3])}, "rectangles_baseline": int(member[0].sum()), "distinct_rectangles_over_variants": len(union),
               "rectangles_in_every_variant": int(member.all(0).sum()), "variants": {}}
        for vi, v in enumerate(VARIANTS):
            row["variants"][v] = {"rect_list_equals_baseline": bool(np.array_equal(member[vi], member[0])), "rect_list_within_1e-4_of_baseline": within_tolerance(runs[vi]["rects"], runs[0]["rects"]),
                                  "rectangles": int(member[vi].sum()),
                                  "segment_records": int(segs[vi, 0]), "segment_records_differing": int(segs[vi, 1]), "segment_max_coordinate_difference": float(segs[vi, 2]),
                                  "pixels_differing": {p: int(planes[vi, pi]) for pi, p in enumerate(PLANES) if planes[vi, pi] != 0}}
        summary["frames"].append(row)
        print(json.dumps(row), flush=True)
    np.savez_compressed(os.path.join(helpers.GOLDEN, "builtin_sensitivity.npz"), **out)
    with open(os.path.join(helpers.GOLDEN, "builtin_sensitivity.json"), "w") as f:
        json.dump(summary, f, indent=1)


if __name__ == "__main__":
    main()
