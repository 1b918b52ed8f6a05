#!/usr/bin/env python3
"""Golden vectors for the individual oclimgutil.h operators, produced by THE REFERENCE (oracle/_ref/librdref.so: its
kernels and its host C on the serial OpenCL shim), written to tests/golden/ops_<w>x<h>.npz together with the inputs.
Run in the container that has /root/reference (after `make -C oracle ref`); the fixtures travel, the reference does not.
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import helpers  # noqa: E402

vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float


class Ops:
    """the reference's (or any ABI-compatible library's) operators behind a tiny numpy front end"""

    def __init__(self, L):
        self.L = L
        for name, res, args in [("simpleGetDevice", vp, [ci]), ("simpleCreateContext", vp, [vp]), ("clCreateCommandQueue", vp, [vp, vp, ctypes.c_ulong, vp]),
                                ("clCreateBuffer", vp, [vp, ctypes.c_ulong, ctypes.c_size_t, vp, vp]), ("clEnqueueReadBuffer", ci, [vp, vp, ci, ctypes.c_size_t, ctypes.c_size_t, vp, ci, vp, vp]),
                                ("clReleaseMemObject", ci, [vp]), ("clFinish", ci, [vp]), ("init_oclimgutil", vp, [vp, vp])]:
            f = getattr(L, name); f.restype = res; f.argtypes = args
        self.dev = L.simpleGetDevice(0)
        self.ctx = L.simpleCreateContext(self.dev)
        self.q = L.clCreateCommandQueue(self.ctx, self.dev, 0, None)
        self.iu = L.init_oclimgutil(self.dev, self.ctx)

    def buf(self, a):
        if isinstance(a, int):
            return self.L.clCreateBuffer(self.ctx, 1, a, None, None)
        a = np.ascontiguousarray(a)
        return self.L.clCreateBuffer(self.ctx, 1 | 32, a.nbytes, a.ctypes.data, None)   # READ_WRITE | COPY_HOST_PTR

    def read(self, m, dtype, count):
        out = np.empty(count, dtype)
        assert self.L.clEnqueueReadBuffer(self.q, m, 1, 0, out.nbytes, out.ctypes.data, 0, None, None) == 0
        return out

    def call(self, name, argtypes, *args):
        f = getattr(self.L, "oclimgutil_" + name)
        f.restype = vp
        f.argtypes = [vp] + argtypes + [vp, vp]
        f(self.iu, *args, self.q, None)
        self.L.clFinish(self.q)


def inputs(iw, ih, seed):
    rng = np.random.default_rng(seed)
    bgr = rng.integers(0, 256, (ih, iw, 3), dtype=np.uint8)
    # a smooth-ish float plane with structure (blobs + noise), positive
    yy, xx = np.mgrid[0:ih, 0:iw].astype(np.float32)
    f = (0.5 + 0.4 * np.sin(xx * 0.21) * np.cos(yy * 0.17) + 0.1 * rng.random((ih, iw))).astype(np.float32)
    plab = rng.integers(0, 2 ** 32, (ih, iw), dtype=np.uint32)
    lab = rng.integers(-3, 40, (ih, iw), dtype=np.int32)
    return bgr, f, plab, lab


def run_all(o, iw, ih, seed):
    """every operator on the same seeded inputs; returns {name: output array}"""
    bgr, f, plab, lab = inputs(iw, ih, seed)
    N = iw * ih
    ws = iw * 3 + 1
    res = {"in_bgr": bgr, "in_f": f, "in_plab": plab, "in_lab": lab}
    mf, mp, ml = o.buf(f), o.buf(plab), o.buf(lab)
    padded = np.zeros((ih, ws), np.uint8); padded[:, : iw * 3] = bgr.reshape(ih, iw * 3)
    mb = o.buf(padded)

    def out_bgr(name, argtypes, *args):
        m = o.buf(np.zeros(ih * ws, np.uint8))
        o.call(name, argtypes, m, *args)
        res[name] = o.read(m, np.uint8, ih * ws).reshape(ih, ws)[:, : iw * 3].copy()
        o.L.clReleaseMemObject(m)

    out_bgr("convert_bgr_lumaf", [vp, vp, cf, ci, ci, ci], mf, 0.9, iw, ih, ws)
    out_bgr("convert_bgr_labeli", [vp, vp, ci, ci, ci, ci], ml, -1, iw, ih, ws)
    out_bgr("convert_bgr_plab", [vp, vp, ci, ci, ci], mp, iw, ih, ws)

    def out_plane(name, argtypes, dtype, count, *args):
        m = o.buf(count * np.dtype(dtype).itemsize)
        o.call(name, argtypes, m, *args)
        res[name] = o.read(m, dtype, count)
        o.L.clReleaseMemObject(m)
        return res[name]

    out_plane("edge_f_f", [vp, vp, ci, ci], np.float32, N, mf, iw, ih)
    vplab = out_plane("edgevec_f2_plab", [vp, vp, ci, ci], np.float32, 2 * N, mp, iw, ih)
    vf = out_plane("edgevec_f2_f", [vp, vp, ci, ci], np.float32, 2 * N, mf, iw, ih)
    mv = o.buf(vf)
    out_plane("thincubic_f_f_f2", [vp, vp, vp, ci, ci], np.float32, N, mf, mv, iw, ih)
    out_plane("thinthres_f_f_f2", [vp, vp, vp, ci, ci], np.float32, N, mf, mv, iw, ih)
    out_plane("edge_f_plab", [vp, vp, ci, ci], np.float32, N, mp, iw, ih)
    out_plane("convert_plab_bgr", [vp, vp, ci, ci, ci], np.uint32, N, mb, iw, ih, ws)
    t0, t1 = o.buf(N * 4), o.buf(N * 4)
    out_plane("iirblur_f_f", [vp, vp, vp, vp, ci, ci, ci], np.float32, N, mf, t0, t1, 2, iw, ih)
    out_plane("threshold_f_f", [vp, vp, cf, cf, cf, ci], np.float32, N, mf, -1.0, 0.55, 2.0, N)
    out_plane("threshold_i_i", [vp, vp, ci, ci, ci, ci], np.int32, N, ml, 7, 10, 9, N)
    out_plane("cast_i_f", [vp, vp, cf, ci], np.int32, N, mf, 1000.0, N)
    res["cast_c_i"] = out_plane("cast_c_i", [vp, vp, ci], np.int8, N, ml, N)
    m0, m1, m2 = o.buf(N * 4), o.buf(N * 4), o.buf(N * 4)
    o.call("unpack_f_f_f_plab", [vp, vp, vp, vp, ci, ci], m0, m1, m2, mp, iw, ih)
    res["unpack0"], res["unpack1"], res["unpack2"] = (o.read(m, np.float32, N) for m in (m0, m1, m2))
    out_plane("pack_plab_f_f_f", [vp, vp, vp, vp, ci, ci], np.uint32, N, m0, m1, m2, iw, ih)
    # strength sums on top of a non-zero accumulator, then the filter
    acc = np.arange(N, dtype=np.int32) % 3
    ms = o.buf(acc)
    o.call("calcStrength", [vp, vp, vp, ci, ci], ms, mf, ml, iw, ih)
    res["calcStrength"] = o.read(ms, np.int32, N)
    ml2 = o.buf(lab)
    o.call("filterStrength", [vp, vp, ci, ci, ci], ml2, ms, 5000, iw, ih)
    res["filterStrength"] = o.read(ml2, np.int32, N)
    for m in (mf, mp, ml, mb, mv, t0, t1, m0, m1, m2, ms, ml2):
        o.L.clReleaseMemObject(m)
    return res


IIR_RADII = [0, 1, 2, 3, 4, 7, 12, 20, 31]


def run_iir_radii(o, iw, ih, seed):
    """oclimgutil_iirblur_f_f for radii no application passes (oclimgutil.c:243-273 with iircoef[r], oclimgutil.cl:900)"""
    _, f, _, _ = inputs(iw, ih, seed)
    N = iw * ih
    res = {"in_f": f}
    for r in IIR_RADII:
        mf, t0, t1, m = o.buf(f), o.buf(N * 4), o.buf(N * 4), o.buf(N * 4)
        o.call("iirblur_f_f", [vp, vp, vp, vp, ci, ci, ci], m, mf, t0, t1, r, iw, ih)
        res["r%d" % r] = o.read(m, np.float32, N)
        for b in (mf, t0, t1, m):
            o.L.clReleaseMemObject(b)
    return res


def main():
    if not helpers.have_ref():
        raise SystemExit("oracle/_ref/librdref.so missing: run `make -C oracle ref` where /root/reference exists")
    o = Ops(ctypes.CDLL(helpers.REF_SO))
    for iw, ih, seed in [(97, 61, 1), (160, 131, 2)]:
        res = run_all(o, iw, ih, seed)
        np.savez_compressed(os.path.join(helpers.GOLDEN, "ops_%dx%d.npz" % (iw, ih)), iw=iw, ih=ih, seed=seed, **res)
        print("ops_%dx%d.npz:" % (iw, ih), ", ".join(sorted(k for k in res if not k.startswith("in_"))))
    for iw, ih, seed in [(97, 61, 3), (160, 131, 4)]:
        res = run_iir_radii(o, iw, ih, seed)
        np.savez_compressed(os.path.join(helpers.GOLDEN, "ops_iir_%dx%d.npz" % (iw, ih)), iw=iw, ih=ih, seed=seed, radii=np.array(IIR_RADII), **res)
        print("ops_iir_%dx%d.npz:" % (iw, ih), ", ".join(k for k in res if k != "in_f"))


if __name__ == "__main__":
    main()
