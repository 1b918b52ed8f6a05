#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  The reference LAUNCH BY LAUNCH on the box's real OpenCL device against the same reference on the serial stand-in - the generator of every golden.

Both sides are the reference's unchanged host C and .cl sources (oracle/_ref/librdref_ocl.so: the system's OpenCL loader behind the observer refshim/rdcl_observe.c;
oracle/_ref/librdref.so: the serial stand-in refshim/rdcl_device.c).  Both start from zeroed buffers (the stand-in's always are; the observer fills the device's on request -
the reference reads planes it never wrote) and see the same frames.  After EVERY one of a frame's 220 launches each buffer argument is fingerprinted on both
sides; at check points of the polyline stage the per-pixel id plane and the segment list are kept and compared UP TO THE PERMUTATION OF IDS that `relabel_pass0` and
`mkpl_pass2` hand out through atomic counters in work-item order: a bijection between the two sides' ids is read off the id planes, the records are
compared field by field through it (pointers and polyline ids mapped).  The question it answers: up to which launch does a real parallel device compute exactly what the
goldens' generator computes, and what is the first launch after which a canonicalised record differs (SURVEY.md predicts `refine_pass3`, H15, and nothing before it).

usage (GPU box):  AMD_OCL_BUILD_OPTIONS_APPEND=<the goldens' contract, tools/gpu_probe_ocl6.sh> python tools/ref_launches_on_opencl.py [tag]
                  -> gpurun_out/ref_launches_opencl_<tag>.json
       (here, no device): RD_LAUNCHES_SELFTEST=1 python tools/ref_launches_on_opencl.py selftest
                  - "device" = the stand-in again with the id-assigning kernels in reversed work-item order: exercises the canonical comparison
internal:         python tools/ref_launches_on_opencl.py --side standin|device|permuted <stream index> <out.npz>"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rectdetect_amd import LS_DTYPE, RECT_DTYPE, synth  # noqa: E402

TAN36 = float(np.tan(36.0 / 180.0 * np.pi))
# (seed offset, width, height, frames reported on; the stream runs from frame 0 so that the state carried from frame to frame is the real one)
STREAMS = [(0, 640, 480, [0]), (5, 640, 480, [0, 1]), (1, 1280, 720, [0]), (0, 1920, 1080, [0, 1]), (12, 1920, 1080, [0, 1, 2])]
MKPL_ROUNDS = [0, 1, 2, 4, 8, 14]
# check points: name -> (kernel, occurrence, argument index)
CHECKPOINTS = {"ids_relabel": ("polyline:relabel_pass1", 0, 0), "ls_pass0b": ("polyline:mkpl_pass0b", 0, 0),
               "ids_final": ("polyline:refine_pass1", 0, 2), "ls_refine2": ("polyline:refine_pass2", 0, 1), "ls_refine3": ("polyline:refine_pass3", 0, 0)}
for _i in MKPL_ROUNDS:
    CHECKPOINTS["ls_split%d" % _i] = ("polyline:mkpl_pass3", _i, 0)
    CHECKPOINTS["ids_split%d" % _i] = ("polyline:mkpl_pass3", _i, 3)
PERMUTED = b"polyline:relabel_pass0,polyline:mkpl_pass2"      # (self test: the kernels that hand out ids, run in reversed raster order)


def load(side):
    R = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "librdref_ocl.so" if side == "device" else "librdref.so"))
    R.rdref_rect_open.restype = ctypes.c_void_p
    R.rdref_rect_open.argtypes = [ctypes.c_int, ctypes.c_int]
    R.rdref_rect_close.argtypes = [ctypes.c_void_p]
    R.rdref_rect_execute_once.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_int]
    R.rdcl_snapshot_request.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    R.rdcl_snapshot_fetch.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_int)]
    R.rdcl_snapshot_limit.argtypes = [ctypes.c_size_t]
    R.rdcl_trace_name.restype = ctypes.c_char_p
    R.rdcl_trace_name.argtypes = [ctypes.c_int]
    R.rdcl_trace_hash.restype = ctypes.c_uint64
    R.rdcl_trace_hash.argtypes = [ctypes.c_int, ctypes.c_int]
    R.rdcl_trace_bytes.restype = ctypes.c_size_t
    R.rdcl_trace_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
    return R


def run_side(side, si, out_path):
    seed, iw, ih, frames = STREAMS[si]
    N = iw * ih
    R = load(side)
    R.rdcl_zero_fill(1)
    if side == "permuted":
        R.rdcl_set_order.argtypes = [ctypes.c_char_p] + [ctypes.c_int] * 4
        R.rdcl_set_order(PERMUTED, 0, 0, 4, 0)
    h = R.rdref_rect_open(iw, ih)
    out = {}
    for t in range(max(frames) + 1):
        img = np.ascontiguousarray(synth.frame(synth.SEED0 + seed, iw, ih, t)).copy()
        R.rdcl_trace_reset()
        R.rdcl_snapshot_clear()
        keep = t in frames
        R.rdcl_hash_all(1 if keep else 0)
        R.rdcl_snapshot_limit(N * 4)
        hs = {k: R.rdcl_snapshot_request(v[0].encode(), v[1], v[2]) for k, v in CHECKPOINTS.items()} if keep else {}
        rects = np.zeros(1024, RECT_DTYPE)
        n = R.rdref_rect_execute_once(h, img.ctypes.data, img.strides[0], TAN36, rects.ctypes.data, 1024)
        if not keep:
            continue
        L = R.rdcl_trace_count()
        out["f%d_names" % t] = np.array([R.rdcl_trace_name(i).decode() for i in range(L)])
        out["f%d_hash" % t] = np.array([[R.rdcl_trace_hash(i, a) for a in range(16)] for i in range(L)], np.uint64)
        out["f%d_bytes" % t] = np.array([[R.rdcl_trace_bytes(i, a) for a in range(16)] for i in range(L)], np.uint64)
        out["f%d_rects" % t] = rects[1:max(1, n)].copy()
        for k, hd in hs.items():
            p, sz, o = ctypes.c_void_p(), ctypes.c_size_t(), ctypes.c_int()
            if R.rdcl_snapshot_fetch(hd, ctypes.byref(p), ctypes.byref(sz), ctypes.byref(o)) != 0:
                continue
            a = np.frombuffer((ctypes.c_char * sz.value).from_address(p.value), dtype="i4").copy()
            if k.startswith("ls_"):
                a = a[: 14 * (min(int(a[0]), len(a) // 14 - 1) + 1)]      # the records in use (record 0 = header: the count)
            out["f%d_%s" % (t, k)] = a
    R.rdref_rect_close(h)
    np.savez_compressed(out_path, **out)


# ---------------------------------------------------------------- comparison up to the permutation of ids

PLAIN = ["x0", "y0", "x1", "y1", "startIndex", "endIndex", "startCount", "endCount", "maxDist", "npix", "level"]
MAPPED = ["leftPtr", "rightPtr", "polyid"]


def id_bijection(ids_d, ids_r, nmax):
    """device id -> stand-in id, read off the two id planes; (pi or None, why)"""
    if ids_d.shape != ids_r.shape:
        return None, "planes of different size"
    if not np.array_equal(ids_d != 0, ids_r != 0):
        return None, "%d pixels carry an id on one side only" % int(((ids_d != 0) != (ids_r != 0)).sum())
    m = ids_d != 0
    pairs = np.unique(np.stack([ids_d[m].astype(np.int64), ids_r[m].astype(np.int64)], 1), axis=0)
    if len(pairs) and (pairs.min() < 0 or pairs.max() > nmax):
        return None, "id outside the list"
    if len(np.unique(pairs[:, 0])) != len(pairs) or len(np.unique(pairs[:, 1])) != len(pairs):
        return None, "the two sides partition the pixels differently (%d id pairs for %d / %d ids)" % (len(pairs), len(np.unique(pairs[:, 0])), len(np.unique(pairs[:, 1])))
    pi = np.full(nmax + 1, -1, np.int64)
    pi[0] = 0
    if len(pairs):
        pi[pairs[:, 0]] = pairs[:, 1]
    return pi, ""


def compare_lists(ids_d, ls_d, ids_r, ls_r):
    """segment lists of the two sides through the id bijection of their id planes"""
    nd, nr = int(ls_d[0]), int(ls_r[0])
    res = {"records": [nd, nr], "identical_as_they_lie": bool(nd == nr and np.array_equal(ls_d, ls_r) and np.array_equal(ids_d, ids_r))}
    if nd != nr:
        res["equal_up_to_ids"] = False
        res["why"] = "record counts differ"
        return res
    pi, why = id_bijection(ids_d, ids_r, nd)
    if pi is None:
        res["equal_up_to_ids"] = False
        res["why"] = why
        return res
    D, Rr = ls_d[: 14 * (nd + 1)].view(LS_DTYPE), ls_r[: 14 * (nr + 1)].view(LS_DTYPE)
    # records no pixel points to are reached through their neighbours' pointers
    for _ in range(4):
        for f in ("leftPtr", "rightPtr"):
            src = np.nonzero(pi[1:] >= 0)[0] + 1
            tgt_d, tgt_r = D[f][src], Rr[f][pi[src]]
            ok = (tgt_d > 0) & (tgt_d <= nd) & (tgt_r > 0) & (tgt_r <= nr)
            new = ok & (pi[np.clip(tgt_d, 0, nd)] < 0)
            pi[tgt_d[new]] = tgt_r[new]
    g = np.nonzero((pi[1:] >= 0) & (D["polyid"][1:] != 0))[0] + 1
    res["records_mapped"] = int(len(g))
    res["valid_records"] = [int((D["polyid"][1:] != 0).sum()), int((Rr["polyid"][1:] != 0).sum())]
    res["ids_permuted"] = bool(len(g) and not np.array_equal(pi[g], g))
    bad = np.zeros(len(g), bool)
    fields = {}
    for f in PLAIN:
        d = D[f][g].view("u4") != Rr[f][pi[g]].view("u4")
        if d.any():
            fields[f] = int(d.sum())
        bad |= d
    for f in MAPPED:
        v = D[f][g]
        inr = (v >= 0) & (v <= nd)
        d = ~inr | (pi[np.clip(v, 0, nd)] != Rr[f][pi[g]])
        if d.any():
            fields[f] = int(d.sum())
        bad |= d
    res["records_differing"] = int(bad.sum())
    if fields:
        res["fields_differing"] = fields
        e = np.abs(np.stack([D[f][g][bad].astype(np.float64) - Rr[f][pi[g]][bad] for f in ("x0", "y0", "x1", "y1")]))
        res["largest_end_point_difference_px"] = float(e.max()) if e.size else 0.0
    res["equal_up_to_ids"] = bool(res["valid_records"][0] == res["valid_records"][1] == len(g) and not bad.any())
    return res


def compare_frame(zd, zr, t):
    names = [str(x) for x in zr["f%d_names" % t]]
    rep = {"launches": len(names)}
    if [str(x) for x in zd["f%d_names" % t]] != names:
        rep["launch_sequences_differ"] = True
        return rep
    hd, hr, by = zd["f%d_hash" % t], zr["f%d_hash" % t], zr["f%d_bytes" % t]
    occ, seen = [], {}
    for nme in names:
        occ.append(seen.get(nme, 0))
        seen[nme] = occ[-1] + 1
    differing = [(i, [int(a) for a in np.nonzero((hd[i] != hr[i]) & (by[i] > 0))[0]]) for i in range(len(names)) if ((hd[i] != hr[i]) & (by[i] > 0)).any()]
    rep["launches_with_every_buffer_argument_identical"] = len(names) - len(differing)
    rep["buffer_arguments_fingerprinted"] = int((by > 0).sum())
    if differing:
        i0 = differing[0][0]
        rep["first_launch_with_a_differing_buffer"] = {"ordinal": i0, "kernel": names[i0], "occurrence": occ[i0], "arguments": differing[0][1]}
        rep["identical_up_to_launch"] = "%s #%d" % (names[i0 - 1], occ[i0 - 1]) if i0 else ""
    per_prog = {}
    for i, nme in enumerate(names):
        p = per_prog.setdefault(nme.split(":")[0], [0, 0])
        p[0] += 1
        p[1] += not any(i == d[0] for d in differing)
    rep["identical_launches_per_program"] = {k: "%d of %d" % (v[1], v[0]) for k, v in per_prog.items()}
    rep["differing_launches"] = ["%s #%d args %s" % (names[i], occ[i], a) for i, a in differing][:40]
    # the polyline stage's check points through the id bijection
    cps = {}
    have = lambda k: ("f%d_%s" % (t, k)) in zd.files and ("f%d_%s" % (t, k)) in zr.files
    get = lambda z, k: z["f%d_%s" % (t, k)]
    if have("ids_relabel"):
        a, b = get(zd, "ids_relabel"), get(zr, "ids_relabel")
        pi, why = id_bijection(a, b, int(max(a.max(), b.max())))
        cps["relabel_pass1 (ids of the chains)"] = {"identical_as_they_lie": bool(np.array_equal(a, b)), "equal_up_to_ids": pi is not None, **({"why": why} if pi is None else {"chains": int(len(np.unique(b)) - 1)})}
        if have("ls_pass0b"):
            cps["mkpl_pass0b (initial segments)"] = compare_lists(a, get(zd, "ls_pass0b"), b, get(zr, "ls_pass0b"))
    for i in MKPL_ROUNDS:
        if have("ls_split%d" % i) and have("ids_split%d" % i):
            cps["mkpl_pass3 #%d (after split round %d)" % (i, i + 1)] = compare_lists(get(zd, "ids_split%d" % i), get(zd, "ls_split%d" % i), get(zr, "ids_split%d" % i), get(zr, "ls_split%d" % i))
    for k, title in (("ls_refine2", "refine_pass2 (end points fitted to the pixels)"), ("ls_refine3", "refine_pass3 (neighbouring segments joined, in place: H15)")):
        if have(k) and have("ids_final"):
            cps[title] = compare_lists(get(zd, "ids_final"), get(zd, k), get(zr, "ids_final"), get(zr, k))
    rep["polyline_check_points"] = cps
    first = [k for k, v in cps.items() if not v.get("equal_up_to_ids", False)]
    rep["first_check_point_that_differs_up_to_ids"] = first[0] if first else None
    key = lambda r: r["c2"].tobytes() + r["c3"].tobytes() + r["value"].tobytes() + r["status"].tobytes()
    rd, rr = zd["f%d_rects" % t].view(RECT_DTYPE), zr["f%d_rects" % t].view(RECT_DTYPE)
    rep["rectangles"] = {"device": len(rd), "stand_in": len(rr), "identical_lists": bool(len(rd) == len(rr) and all(key(a) == key(b) for a, b in zip(rd, rr))),
                         "identical_sets": set(key(a) for a in rd) == set(key(b) for b in rr)}
    return rep


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--side":
        run_side(sys.argv[2], int(sys.argv[3]), sys.argv[4])
        return
    tag = sys.argv[1] if len(sys.argv) > 1 else "default"
    selftest = os.environ.get("RD_LAUNCHES_SELFTEST") == "1"
    which = [int(x) for x in os.environ["RD_LAUNCHES_STREAMS"].split(",")] if os.environ.get("RD_LAUNCHES_STREAMS") else range(len(STREAMS))
    rep = {"build_options_appended": os.environ.get("AMD_OCL_BUILD_OPTIONS_APPEND", ""), "device_side": "the stand-in with the id-assigning kernels in reversed order (self test)" if selftest else "the box's OpenCL device", "frames": {}}
    tmp = os.environ.get("TMPDIR", "/tmp")
    for si in which:
        seed, iw, ih, frames = STREAMS[si]
        pd, pr = os.path.join(tmp, "rl_device_%d.npz" % si), os.path.join(tmp, "rl_standin_%d.npz" % si)
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--side", "permuted" if selftest else "device", str(si), pd]),
                 subprocess.Popen([sys.executable, os.path.abspath(__file__), "--side", "standin", str(si), pr])]
        rcs = [p.wait() for p in procs]
        if any(rcs):
            rep["frames"]["seed %d %dx%d" % (seed, iw, ih)] = {"error": "a side failed: exit codes %s" % rcs}
            continue
        zd, zr = np.load(pd), np.load(pr)
        for t in frames:
            name = "seed %d %dx%d t %d" % (seed, iw, ih, t)
            r = rep["frames"][name] = compare_frame(zd, zr, t)
            print(name, "| identical launches:", r.get("identical_launches_per_program"), "| first differing buffer:", (r.get("first_launch_with_a_differing_buffer") or {}).get("kernel"),
                  "| polyline check points equal up to ids:", {k.split(" ")[0] + (k.split(" ")[1] if "#" in k else ""): v.get("equal_up_to_ids") for k, v in r.get("polyline_check_points", {}).items()},
                  "| first that differs:", r.get("first_check_point_that_differs_up_to_ids"), "| rectangles", r.get("rectangles"), flush=True)
        zd.close()
        zr.close()
        os.remove(pd)
        os.remove(pr)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rep, open(os.path.join(ROOT, "gpurun_out", "ref_launches_opencl_%s.json" % tag), "w"), indent=1)


if __name__ == "__main__":
    main()
