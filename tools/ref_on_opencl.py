#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  The reference on a REAL OpenCL device: `oracle/_ref/librdref_ocl.so` is the reference's unchanged host C
(oracle/Makefile: ref_ocl) linked against the system's OpenCL loader, so on the GPU box the reference compiles its own .cl sources with ROCm's OpenCL
compiler (default options: contraction ON, the device's builtins) and runs them on the MI355X in whatever work-item order the device takes.  This is the only
third-party execution of the reference this environment offers; it bounds what the "unpinned" oracle (the reference on our serial stand-in, built with
-ffp-contract=off) hides:
  * per fixture frame, whether the rectangle list equals the golden's bit for bit / within the 1e-4 bar, and which of the stand-in's builtin variants
    (tests/golden/builtin_sensitivity.npz: baseline, rsqrt1, hypotf, distd, fma, all) it equals;
  * the polyline fixture's segment list against the golden;
  * run-to-run repeatability of the reference on this device (the in-place region stages depend on the work-item order);
  * the reference's frames/s on this device (oclrect_enqueueTask / pollTask two deep as vidrect.cpp does, and oclrect_executeOnce) - a same-box baseline.
usage (GPU box): python tools/ref_on_opencl.py [section ...]   -> gpurun_out/ref_opencl.json     (sections: stills poly stream repeat timing)
Build options can be appended without touching the reference through the AMD runtime's AMD_OCL_BUILD_OPTIONS_APPEND, e.g. "-Wf,-ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt"
(the arithmetic contract of the goldens): tools/gpu_probe_ocl2.sh."""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rectdetect_amd import LS_DTYPE, RECT_DTYPE, synth  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
SO = os.path.join(ROOT, "oracle", "_ref", "librdref_ocl.so")
TAN36 = float(np.tan(36.0 / 180.0 * np.pi))
OUT = os.path.join(ROOT, "gpurun_out", "ref_opencl.json")


def load():
    R = ctypes.CDLL(SO)
    R.rdref_rect_open.restype = ctypes.c_void_p
    R.rdref_rect_open.argtypes = [ctypes.c_int, ctypes.c_int]
    R.rdref_rect_close.argtypes = [ctypes.c_void_p]
    R.rdref_rect_execute_once.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_int]
    R.rdref_rect_enqueue.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    R.rdref_rect_poll.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_int]
    R.rdref_poly_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    return R


def rect_key(r):
    return r["c2"].tobytes() + r["c3"].tobytes() + r["value"].tobytes() + r["status"].tobytes()


def compare(a, b, tol=1e-4):
    """a, b: rect lists -> (equal as sets bit for bit, within the north_star's bar as sets, rectangles of a in b bit for bit, largest coordinate distance
    of a's rectangles to their nearest in b by the four image corners)"""
    ka, kb = {rect_key(r) for r in a}, {rect_key(r) for r in b}
    within = False
    if len(a) == len(b):
        if len(a) == 0:
            within = True
        else:
            canon = lambda rs: rs[np.lexsort(np.rint(rs["c2"]).reshape(len(rs), 8).T[::-1])]
            x, y = canon(a), canon(b)
            within = bool(np.array_equal(x["status"], y["status"]) and np.array_equal(np.rint(x["c2"]), np.rint(y["c2"])) and np.abs(x["c2"] - y["c2"]).max() <= tol and
                          np.abs(x["c3"] - y["c3"]).max() <= tol and np.abs(x["value"] - y["value"]).max() <= tol)
    far = 0.0
    matched = 0
    for r in a:
        if len(b) == 0:
            far = float("inf")
            break
        d = np.abs(b["c2"] - r["c2"]).reshape(len(b), -1).max(1)
        far = max(far, float(d.min()))
        matched += int(d.min() <= 1.0)
    return {"equal": ka == kb, "within_1e-4": within, "n": len(a), "n_golden": len(b), "n_distinct": len(ka), "n_golden_distinct": len(kb), "common_bitwise": len(ka & kb), "matched_within_1px": matched, "max_nearest_corner_distance": far}


class Rect:
    def __init__(self, R, iw, ih):
        self.R, self.h = R, R.rdref_rect_open(iw, ih)

    def once(self, bgr, tan=TAN36):
        a = np.ascontiguousarray(bgr).copy()
        out = np.zeros(1024, RECT_DTYPE)
        k = self.R.rdref_rect_execute_once(self.h, a.ctypes.data, a.strides[0], float(tan), out.ctypes.data, 1024)
        return out[1:k].copy()

    def close(self):
        self.R.rdref_rect_close(self.h)


def section_stills(R, rep):
    z = np.load(os.path.join(GOLDEN, "builtin_sensitivity.npz"))
    variants = [str(v) for v in z["variants"]]
    rows = []
    for fi, (seed, iw, ih, t) in enumerate(z["frames"].tolist()):
        d = Rect(R, iw, ih)
        got = d.once(synth.frame(synth.SEED0 + seed, iw, ih, t))
        d.close()
        union, member = z["f%d_union" % fi], z["f%d_member" % fi]
        row = {"frame": {"seed": seed, "iw": iw, "ih": ih, "t": t}}
        for vi, v in enumerate(variants):
            row[v] = compare(got, union[member[vi] != 0])
        row["equals_variants"] = [v for v in variants if row[v]["equal"]]
        row["within_1e-4_of_variants"] = [v for v in variants if row[v]["within_1e-4"]]
        rows.append(row)
        print("still", row["frame"], "n", len(got), "equals", row["equals_variants"], "within", row["within_1e-4_of_variants"], "baseline", row["baseline"], flush=True)
    rep["stills"] = rows


def section_poly(R, rep):
    rows = []
    for name in ("poly_640x480_s0", "poly_333x217_s2"):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        iw, ih = int(z["iw"]), int(z["ih"])
        bgr = np.ascontiguousarray(synth.frame(int(z["seed"]), iw, ih, 0))
        N = iw * ih
        ls = np.zeros(N * 4, np.int32)
        ids = np.zeros(N, np.int32)
        n = R.rdref_poly_run(bgr.ctypes.data, iw, ih, bgr.strides[0], int(z["strength_thre"]), float(z["minerror"]), int(z["size_thre"]), ls.ctypes.data, ids.ctypes.data, None)
        got, gold = ls[: 14 * (n + 1)].view(LS_DTYPE), z["segments"]
        ng = int(gold.view("i4")[0])
        row = {"fixture": name, "segments": int(n), "segments_golden": ng}
        if n == ng:
            m = (got[1:]["polyid"] != 0) & (gold[1:]["polyid"] != 0)
            row["records_differing"] = int(sum(got[1:][m][k].tobytes() != gold[1:][m][k].tobytes() for k in range(int(m.sum()))))
            row["max_coordinate_difference"] = max(float(np.abs(got[1:][m][c] - gold[1:][m][c]).max(initial=0)) for c in ("x0", "y0", "x1", "y1"))
        # (record ids follow the work-item order of the split rounds: compare the segments as a set of end-point quadruples as well)
        quad = lambda a: {(float(r["x0"]), float(r["y0"]), float(r["x1"]), float(r["y1"])) for r in a[1:] if r["polyid"] != 0}
        qa, qb = quad(got), quad(gold)
        row["valid_segments"], row["valid_segments_golden"], row["segments_common_as_end_point_quadruples"] = len(qa), len(qb), len(qa & qb)
        # the golden segments that have no identical twin: how far is the nearest segment of this run (largest end-point coordinate difference, either direction of travel)?
        A = np.array(sorted(qa), np.float64).reshape(-1, 4)
        near = []
        for g in sorted(qb - qa):
            g = np.array(g)
            d = np.minimum(np.abs(A - g).max(1), np.abs(A - g[[2, 3, 0, 1]]).max(1)).min() if len(A) else float("inf")
            near.append(float(d))
        row["unmatched_golden_segments_nearest_distance"] = sorted(round(d, 3) for d in near)
        rows.append(row)
        print("poly", row, flush=True)
    rep["poly"] = rows


def section_stream(R, rep):
    rows = {}
    for name, nmax in (("stream_1920x1080_s0", 16), ("stream_1280x720_s1", 30)):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        iw, ih, seed, tan = int(z["iw"]), int(z["ih"]), int(z["seed"]), float(z["tan_aov"])
        d = Rect(R, iw, ih)
        fr = []
        for t in range(min(nmax, int(z["nframes"]))):
            c = compare(d.once(synth.frame(seed, iw, ih, t), tan), z["f%d_rects" % t])
            fr.append(c)
        d.close()
        rows[name] = {"frames": len(fr), "equal": sum(c["equal"] for c in fr), "within_1e-4": sum(c["within_1e-4"] for c in fr), "same_count": sum(c["n"] == c["n_golden"] for c in fr),
                      "rectangles": sum(c["n"] for c in fr), "rectangles_golden": sum(c["n_golden"] for c in fr), "rectangles_matched_within_1px": sum(c["matched_within_1px"] for c in fr),
                      "per_frame": fr}
        print("stream", name, {k: v for k, v in rows[name].items() if k != "per_frame"}, flush=True)
    rep["streams"] = rows


def section_streamlong(R, rep):
    """the long streams of the parity table (916 frames: 516 + round 5's two held-out streams), rectangle lists only"""
    rows = {}
    for name in ("stream_1920x1080_s0_100", "stream_1920x1080_s7_100", "stream_1280x720_s1_300", "stream_3840x2160_s4_16", "stream_1920x1080_s11_200", "stream_1920x1080_s12_200"):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        iw, ih, seed, tan = int(z["iw"]), int(z["ih"]), int(z["seed"]), float(z["tan_aov"])
        d = Rect(R, iw, ih)
        fr = [compare(d.once(synth.frame(seed, iw, ih, t), tan), z["f%d_rects" % t]) for t in range(int(z["nframes"]))]
        d.close()
        far = [c["max_nearest_corner_distance"] for c in fr if c["n"]]
        rows[name] = {"frames": len(fr), "lists_bit_identical": sum(c["equal"] for c in fr), "lists_within_1e-4": sum(c["within_1e-4"] for c in fr), "same_count": sum(c["n"] == c["n_golden"] for c in fr),
                      "rectangles_golden_distinct": sum(c["n_golden_distinct"] for c in fr), "rectangles_bit_identical": sum(c["common_bitwise"] for c in fr),
                      "rectangles": sum(c["n"] for c in fr), "rectangles_matched_within_1px": sum(c["matched_within_1px"] for c in fr),
                      "largest_corner_distance_to_nearest_golden_px": max(far) if far else 0.0, "frames_with_a_rectangle_further_than_0.01px": sum(1 for v in far if v > 0.01)}
        print("streamlong", name, rows[name], flush=True)
    rep["streams_long"] = rows


def section_repeat(R, rep):
    """the same frame, fresh detector each time: does this device repeat itself?"""
    iw, ih = 1920, 1080
    lists = []
    for k in range(5):
        d = Rect(R, iw, ih)
        lists.append(d.once(synth.frame(synth.SEED0, iw, ih, 0)))
        d.close()
    rep["repeat_1080p_frame0"] = {"runs": len(lists), "distinct_lists": len({b"".join(sorted(rect_key(r) for r in l)) for l in lists}), "counts": [len(l) for l in lists]}
    print("repeat", rep["repeat_1080p_frame0"], flush=True)


def section_timing(R, rep):
    out = {}
    for iw, ih, n in ((1920, 1080, 60), (1280, 720, 60)):
        frames = [np.ascontiguousarray(synth.frame(synth.SEED0, iw, ih, t)) for t in range(8)]
        d = Rect(R, iw, ih)
        res = np.zeros(1024, RECT_DTYPE)
        for t in range(3):
            d.once(frames[t])
        lat = []
        for t in range(20):
            t0 = time.perf_counter()
            d.once(frames[t % 8])
            lat.append(time.perf_counter() - t0)
        # vidrect.cpp:159-205: a frame is enqueued before the previous one is polled
        R.rdref_rect_enqueue(d.h, frames[0].ctypes.data, frames[0].strides[0])
        t0 = time.perf_counter()
        for t in range(1, n + 1):
            f = frames[t % 8]
            R.rdref_rect_enqueue(d.h, f.ctypes.data, f.strides[0])
            R.rdref_rect_poll(d.h, TAN36, res.ctypes.data, 1024)
        dt = time.perf_counter() - t0
        R.rdref_rect_poll(d.h, TAN36, res.ctypes.data, 1024)
        d.close()
        out["%dx%d" % (iw, ih)] = {"two_deep_frames_per_s": n / dt, "execute_once_ms_median": float(np.median(lat) * 1e3), "frames": n}
        print("timing", iw, ih, out["%dx%d" % (iw, ih)], flush=True)
    rep["timing"] = out


def bench_line():
    """bench.py's baseline leg: the reference on this box's OpenCL device, 1920x1080, one JSON line on stdout (bounded: ~40 frames)"""
    R = load()
    iw, ih, n = 1920, 1080, 40
    frames = [np.ascontiguousarray(synth.frame(synth.SEED0, iw, ih, t)) for t in range(4)]
    t_init = time.perf_counter()
    d = Rect(R, iw, ih)
    res = np.zeros(1024, RECT_DTYPE)
    d.once(frames[0])
    t_init = time.perf_counter() - t_init
    lat = []
    for t in range(8):
        t0 = time.perf_counter()
        d.once(frames[t % 4])
        lat.append(time.perf_counter() - t0)
    R.rdref_rect_enqueue(d.h, frames[0].ctypes.data, frames[0].strides[0])
    t0 = time.perf_counter()
    for t in range(1, n + 1):
        f = frames[t % 4]
        R.rdref_rect_enqueue(d.h, f.ctypes.data, f.strides[0])
        R.rdref_rect_poll(d.h, TAN36, res.ctypes.data, 1024)
    dt = time.perf_counter() - t0
    R.rdref_rect_poll(d.h, TAN36, res.ctypes.data, 1024)
    d.close()
    print(json.dumps({"value": round(n / dt, 2), "unit": "frames/s", "kind": "reference", "device": "this box's OpenCL device (ROCm OpenCL: the GPU the HIP path runs on)",
                      "execute_once_ms_median": round(float(np.median(lat)) * 1e3, 2), "init_and_first_frame_s": round(t_init, 1),
                      "sample": "%d consecutive 1920x1080 frames, oclrect_enqueueTask / oclrect_pollTask two deep (vidrect.cpp:159-205), after 9 untimed; the reference's unchanged host C and .cl sources, "
                                "built at run time by the device's OpenCL compiler with the reference's own (empty) options" % n}), flush=True)


def dump(path, specs):
    """rectangle lists of single frames (seed offset, iw, ih, t), each from a fresh detector -> npz (f0, f1, ..): what tests compare the HIP path with"""
    R = load()
    out = {}
    for i, (seed, iw, ih, t) in enumerate(specs):
        d = Rect(R, iw, ih)
        out["f%d" % i] = d.once(synth.frame(synth.SEED0 + seed, iw, ih, t))
        d.close()
    np.savez(path, **out)


def main():
    if sys.argv[1:] == ["bench"]:
        bench_line()
        return
    if len(sys.argv) > 2 and sys.argv[1] == "dump":
        v = [int(a) for a in sys.argv[3:]]
        dump(sys.argv[2], [tuple(v[k:k + 4]) for k in range(0, len(v), 4)])
        return
    want = sys.argv[1:] or ["stills", "poly", "stream", "repeat", "timing"]
    try:
        with open(OUT) as f:
            rep = json.load(f)
    except (OSError, ValueError):
        rep = {}
    R = load()
    rep["library"] = "oracle/_ref/librdref_ocl.so: the reference's host C unchanged + its .cl sources built at run time by the device's OpenCL compiler with the reference's own (empty) options"
    for s in want:
        globals()["section_" + s](R, rep)
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        with open(OUT, "w") as f:
            json.dump(rep, f, indent=1)


if __name__ == "__main__":
    main()
