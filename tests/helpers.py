"""Test infrastructure: ctypes access to the oracle (oracle/librd_oracle.so, our CPU restatement) and, where it was
built, to the reference itself (oracle/_ref/librdref.so).  Only tests, smoke() and bench.py's cpu_baseline leg use this."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "librd_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "librdref.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")

RECT_PLANES = {  # name -> (dtype, ints per pixel)
    "plab0": ("u4", 1), "plab1": ("u4", 1), "Lblur": ("f4", 1), "vxy": ("f4", 2), "strength": ("f4", 1), "nms": ("f4", 1),
    "mask0": ("i4", 1), "tidy": ("i4", 1), "label1": ("i4", 1), "str_sum": ("i4", 1), "edge500": ("i4", 1), "smooth": ("u4", 1),
    "quant": ("u4", 1), "strong": ("i4", 1), "junction": ("i4", 1), "mergemask": ("i4", 1), "region": ("i4", 1), "rsize": ("i4", 1),
    "boundary_src": ("i4", 1), "boundary": ("i4", 1), "lsid": ("i4", 1), "lslist": ("i4", 4), "table": ("i4", 4),
}

_oracle = None
_ref = None


def P(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], stdout=subprocess.DEVNULL)
        O = ctypes.CDLL(ORACLE_SO)
        O.rdo_rect_new.restype = ctypes.c_void_p
        O.rdo_rect_new.argtypes = [ctypes.c_int, ctypes.c_int]
        O.rdo_rect_free.argtypes = [ctypes.c_void_p]
        O.rdo_rect_plane.restype = ctypes.c_void_p
        O.rdo_rect_plane.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        O.rdo_rect_frame.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        O.rdo_rect_set_region_mode.argtypes = [ctypes.c_void_p, ctypes.c_int]
        O.rdo_rect_info.argtypes = [ctypes.c_void_p, ctypes.c_int]
        O.rdo_region_concurrent.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3
        O.rdo_despeckle2_jacobi_k.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        O.rdo_region_size.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        O.rdo_despeckle2.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        O.rdo_mark_boundary.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        O.rdo_label8.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        O.rdo_reduce_ls.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        O.rdo_poly_frame.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int]
        O.rdo_polyline.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        O.rdo_lut.restype = ctypes.POINTER(ctypes.c_uint16)
        O.rdo_iirblur_r.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        O.rdo_iircoef.restype = ctypes.POINTER(ctypes.c_float)
        O.rdo_iircoef.argtypes = [ctypes.c_int]
        _oracle = O
    return _oracle


REGION_REFERENCE_RASTER = 0       # the reference's in-place region merge, work-items in serial raster order (what oracle/_ref runs by default)
REGION_SPEC = 1                   # the same kernel with the work-items of a launch running concurrently, launched until nothing changes: what the HIP path reproduces bit for bit
REGION_REFERENCE_CONCURRENT = 2   # concurrent work-items, the reference's 8 launches: what oracle/_ref computes under rdcl_set_order(..., 0, 0, 5, 0)
# (the absorption of small regions is the serial raster order's result in every mode - the HIP path evaluates that recurrence exactly)


class OracleRect:
    """All device stages of the rect path on the CPU (oracle/rd_oracle.c), every plane inspectable.
    region_mode selects how the two order-dependent region kernels of the reference are evaluated."""

    def __init__(self, iw, ih, region_mode=REGION_REFERENCE_RASTER):
        self.iw, self.ih, self.N = iw, ih, iw * ih
        self.h = oracle().rdo_rect_new(iw, ih)
        oracle().rdo_rect_set_region_mode(self.h, region_mode)

    def rounds(self):
        """(launches of the merge kernel the last frame evaluated - in REGION_SPEC mode: including the one that changed nothing -, 0)"""
        return oracle().rdo_rect_info(self.h, 1), oracle().rdo_rect_info(self.h, 2)

    def frame(self, bgr):
        a = np.ascontiguousarray(bgr)
        oracle().rdo_rect_frame(self.h, a.ctypes.data, a.strides[0])

    def set_prev_strong(self, strong):
        """the state the next frame inherits: the previous frame's strong-edge mask, e.g. taken from another implementation"""
        a = np.ascontiguousarray(strong, dtype=np.int32)
        assert a.size == self.N
        ctypes.memmove(oracle().rdo_rect_plane(self.h, b"prev_strong"), a.ctypes.data, a.nbytes)

    def plane(self, name):
        dt, k = RECT_PLANES[name]
        p = oracle().rdo_rect_plane(self.h, name.encode())
        return np.frombuffer((ctypes.c_char * (self.N * 4 * k)).from_address(p), dtype=dt).copy()

    def segments(self):
        from rectdetect_amd import LS_DTYPE
        raw = self.plane("lslist")
        n = int(raw[0])
        return raw[: 14 * (n + 1)].view(LS_DTYPE).copy()

    def close(self):
        oracle().rdo_rect_free(self.h)


def oracle_poly(bgr, strength_thre=500, minerror=1.0, size_thre=20):
    from rectdetect_amd import LS_DTYPE
    a = np.ascontiguousarray(bgr)
    ih, iw = a.shape[:2]
    N = iw * ih
    ls = np.zeros(N * 4, np.int32)
    ids = np.zeros(N, np.int32)
    oracle().rdo_poly_frame(P(ls), P(ids), a.ctypes.data, iw, ih, a.strides[0], strength_thre, minerror, size_thre)
    n = int(ls[0])
    return ls[: 14 * (n + 1)].view(LS_DTYPE).copy(), ids


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        R = ctypes.CDLL(REF_SO)
        R.rdref_rect_open.restype = ctypes.c_void_p
        R.rdref_rect_open.argtypes = [ctypes.c_int, ctypes.c_int]
        R.rdref_rect_close.argtypes = [ctypes.c_void_p]
        R.rdref_rect_execute_once.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_int]
        R.rdref_rect_enqueue.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        R.rdref_rect_poll.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_int]
        R.rdref_poly_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        R.rdcl_snapshot_request.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
        R.rdcl_snapshot_fetch.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_int)]
        R.rdcl_trace_name.restype = ctypes.c_char_p
        R.rdcl_trace_name.argtypes = [ctypes.c_int]
        _ref = R
    return _ref


# where each oracle plane lives in a run of the reference: (kernel, occurrence, argument index)
REF_SNAPSHOTS = {
    "plab0": ("imgutil:bgr2plab", 0, 0), "Lblur": ("imgutil:iirblur_f_f_pass3", 2, 0), "plab1": ("imgutil:pack_plab", 0, 0),
    "vxy": ("imgutil:edgevec_f", 0, 0), "strength": ("imgutil:edge_plab", 0, 0), "nms": ("imgutil:thinthres_f_f_f2", 0, 0),
    "mask0": ("imgutil:cast_i_f", 0, 0), "tidy": ("rect:stringify", 1, 0), "str_sum": ("rect:calcStrength", 0, 0),
    "edge500": ("imgutil:threshold_i_i", 0, 0), "smooth": ("rect:blblur1", 9, 0), "quant": ("rect:despeckle", 0, 0),
    "strong": ("imgutil:threshold_i_i", 1, 0), "label1": ("rect:filterStrength", 1, 0), "junction": ("rect:simpleJunction", 1, 0),
    "mergemask": ("rect:mkMergeMask1", 0, 0), "rsize": ("rect:calcSize", 0, 0), "region": ("rect:despeckle2", 0, 0),
    "boundary_src": ("rect:markBoundary", 0, 0), "boundary": ("imgutil:label8xMain_int_int", 19, 0),
    "lslist": ("polyline:refine_pass3", 0, 0), "lsid": ("polyline:refine_pass1", 0, 2), "table": ("rect:reduceLS", 0, 0),
    "flags_label1": ("imgutil:label8xMain_int_int", 9, 2), "flags_boundary": ("imgutil:label8xMain_int_int", 19, 2),
    "flags_chain": ("polyline:label8xMain_int_int", 9, 2), "flags_sub": ("polyline:labelpl_main", 10, 2),
}


def ref_snapshot_fetch(handle, dtype):
    p = ctypes.c_void_p()
    sz = ctypes.c_size_t()
    o = ctypes.c_int()
    if ref().rdcl_snapshot_fetch(handle, ctypes.byref(p), ctypes.byref(sz), ctypes.byref(o)) != 0:
        raise RuntimeError("snapshot not taken")
    return np.frombuffer((ctypes.c_char * sz.value).from_address(p.value), dtype=dtype).copy()


class RefRect:
    """The reference's oclrect_* API running on the serial OpenCL shim, with access to intermediate planes."""

    def __init__(self, iw, ih):
        self.iw, self.ih = iw, ih
        self.h = ref().rdref_rect_open(iw, ih)

    def execute_once(self, bgr, tan_aov, snapshots=()):
        from rectdetect_amd import RECT_DTYPE
        R = ref()
        a = np.ascontiguousarray(bgr).copy()
        R.rdcl_trace_reset()
        R.rdcl_snapshot_clear()
        hs = {k: R.rdcl_snapshot_request(REF_SNAPSHOTS[k][0].encode(), REF_SNAPSHOTS[k][1], REF_SNAPSHOTS[k][2]) for k in snapshots}
        out = np.zeros(1024, RECT_DTYPE)
        k = R.rdref_rect_execute_once(self.h, a.ctypes.data, a.strides[0], float(tan_aov), out.ctypes.data, 1024)
        self.launches = R.rdcl_trace_count()
        snaps = {name: ref_snapshot_fetch(h, "u4") for name, h in hs.items()}
        return out[1:k].copy(), snaps

    def host_postprocess(self, segs, boundary, table, tan_aov):
        """THE REFERENCE'S OWN compiled executeCPUTask (oclrect.c:1049-1226) on the caller's planes: the stand-in hands `segs` (linesegment_t records incl. the header
        record), `table` (reduceLS's vote table) and `boundary` (boundary-component ids, N ints) to the three read-backs of genGPUTask (oclrect.c:371-376) in place of its
        own buffers, and runs none of the 220 launches (rdcl_skip_launches) - what is left of oclrect_executeOnce is the reference's host side, unchanged.  Returns its list."""
        from rectdetect_amd import RECT_DTYPE
        R = ref()
        R.rdcl_substitute_reads.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
        N = self.iw * self.ih
        segs = np.ascontiguousarray(segs).view(np.uint8).reshape(-1)
        ls = np.zeros(N * 16, np.uint8)                  # (the whole read-back: nothing of an earlier frame's list behind the records)
        ls[: len(segs)] = segs
        tb = np.zeros(N * 4, np.int32)
        t = np.ascontiguousarray(table, dtype=np.int32).reshape(-1)
        tb[: len(t)] = t
        bd = np.ascontiguousarray(boundary, dtype=np.int32).reshape(-1)
        assert len(bd) == N
        bufs = [ls, tb, bd]
        ptrs = (ctypes.c_void_p * 3)(*[b.ctypes.data for b in bufs])
        sizes = (ctypes.c_size_t * 3)(*[b.nbytes for b in bufs])
        img = np.zeros((self.ih, self.iw, 3), np.uint8)
        out = np.zeros(4096, RECT_DTYPE)
        R.rdcl_skip_launches(1)
        R.rdcl_substitute_reads(3, ptrs, sizes)
        try:
            k = R.rdref_rect_execute_once(self.h, img.ctypes.data, img.strides[0], float(tan_aov), out.ctypes.data, 4096)
        finally:
            R.rdcl_skip_launches(0)
            R.rdcl_substitute_reads(0, ptrs, sizes)
        assert k <= 4096
        return out[1:k].copy()

    def close(self):
        ref().rdref_rect_close(self.h)


def rects_equal(a, b):
    return len(a) == len(b) and all(a[f].tobytes() == b[f].tobytes() for f in ("c2", "c3", "value", "status"))


def segments_equal(a, b):
    """bit-exact comparison of the valid records of two linesegment_t lists (records with polyid == 0 carry no meaning)."""
    if len(a) != len(b) or int(a.view("i4")[0]) != int(b.view("i4")[0]):
        return False
    va, vb = a[1:], b[1:]
    if not np.array_equal(va["polyid"] != 0, vb["polyid"] != 0):
        return False
    m = va["polyid"] != 0
    return va[m].tobytes() == vb[m].tobytes()


# ---- parity report: what the GPU tests MEASURE about the deviation from the reference's raster-order output (counts, not pass / fail), kept as a
# file so that it shows up in the driver's record (__graft_entry__.smoke() prints it) instead of being swallowed by `pytest -q`.  The tests write the scratch
# copy under gpurun_out/ only (untracked; stamped with the commit it was measured at and cleared when the commit changes); tools/update_parity_report.py
# copies it to the tracked tests/parity_report.json when a round's numbers are to be kept.
PARITY_REPORT = os.path.join(ROOT, "tests", "parity_report.json")
PARITY_SCRATCH = os.path.join(ROOT, "gpurun_out", "parity_report.json")


def _commit():
    import subprocess
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:
        return None


def parity_report(section, key, value):
    import json
    here = _commit() or os.environ.get("RD_COMMIT") or "unknown (no .git on the GPU box)"
    try:
        with open(PARITY_SCRATCH) as f:
            rep = json.load(f)
        if rep.get("_measured_at", {}).get("commit") != here:
            rep = {}
    except (OSError, ValueError):
        rep = {}
    rep["_measured_at"] = {"commit": here}
    rep.setdefault(section, {})[key] = value
    os.makedirs(os.path.dirname(PARITY_SCRATCH), exist_ok=True)
    with open(PARITY_SCRATCH, "w") as f:
        json.dump(rep, f, indent=1, sort_keys=True)
