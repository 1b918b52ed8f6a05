"""rectdetect_amd - MI355X-native rectangle/polyline detector (host-side Python mirror of the C API).

The product is ``librectdetect_hip.so`` (hand-written gfx950 HIP kernels behind the reference's C API:
``oclimgutil.h`` / ``oclpolyline.h`` / ``oclrect.h`` / ``oclhelper.h``).  This module only binds that C ABI with
ctypes so that tests and ``bench.py`` can drive it; it contains no detector logic and NO CPU fallback: if the
library or a GPU is missing, calls fail loudly.

Mirrors of the reference programs' call sequences:
  * :func:`poly_frame`  - poly.cpp:104-131 (explicit operator sequence + ``oclpolyline_execute``)
  * :class:`RectDetector` - rect.cpp / vidrect.cpp (``oclrect_executeOnce`` / ``enqueueTask`` / ``pollTask``)
  * :class:`Detector` - the ``rd_detector`` extension (device-resident frames, several frames in flight)
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RD_LIB_PATH") or os.path.join(_HERE, "librectdetect_hip.so")      # (RD_LIB_PATH: tuning builds, tools/variants.sh)

RECT_DTYPE = np.dtype([("c2", "<f8", (4, 2)), ("c3", "<f8", (4, 3)), ("value", "<f8"), ("status", "<u4"), ("_pad", "<u4")])
LS_DTYPE = np.dtype([("x0", "<f4"), ("y0", "<f4"), ("x1", "<f4"), ("y1", "<f4"), ("startIndex", "<i4"), ("endIndex", "<i4"),
                     ("leftPtr", "<i4"), ("rightPtr", "<i4"), ("startCount", "<i4"), ("endCount", "<i4"), ("maxDist", "<i4"),
                     ("polyid", "<i4"), ("npix", "<i4"), ("level", "<i4")])
assert RECT_DTYPE.itemsize == 176 and LS_DTYPE.itemsize == 56

CL_MEM_READ_WRITE = 1 << 0
CL_MEM_COPY_HOST_PTR = 1 << 5
CL_TRUE = 1

_lib = None


def lib():
    """The loaded C-ABI library; raises if it has not been built (run ``__graft_entry__.build()``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("rectdetect_amd: %s is missing - build it (python -c 'import __graft_entry__ as g; g.build()'). "
                               "There is no CPU fallback." % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def _declare(L):
    vp, ci, cf, cd, cz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_size_t
    sig = {
        "rd_version": (ctypes.c_char_p, []),
        "rd_device_count": (ci, []),
        "rd_select_device": (None, [ci]),
        "rd_device_pci_bus_id": (ci, [ci, ctypes.c_char_p, ci]),
        "rd_device_alloc": (vp, [cz]),
        "rd_device_free": (None, [vp]),
        "rd_host_alloc": (vp, [cz]),
        "rd_host_free": (None, [vp]),
        "rd_release_cached_streams": (None, [ci]),
        "rd_upload": (None, [vp, vp, cz]),
        "rd_download": (None, [vp, vp, cz]),
        "rd_detector_create": (vp, [ci, ci, ci, ci, ci]),
        "rd_detector_destroy": (None, [vp]),
        "rd_detector_enqueue": (ctypes.c_long, [vp, vp, ci, ci]),
        "rd_detector_poll": (vp, [vp, cd]),
        "rd_detector_drain": (None, [vp]),
        "rd_detector_set_aperture": (None, [vp, cd]),
        "rd_detector_counter": (ctypes.c_long, [vp, ci]),
        "rd_detector_last_segments": (ci, [vp, vp, ci]),
        "rd_detector_debug_plane": (cz, [vp, ctypes.c_char_p, vp, cz]),
        "rd_postprocess_planes": (vp, [vp, vp, vp, ci, ci, cd]),
        "rd_post_run": (vp, [vp, ci, vp, ci, ci, cd]),
        "rd_post_helpers_configure": (None, [ci]),
        "rd_post_helpers_arm": (None, []),
        "rd_post_helpers": (ci, []),
        "rd_synth_frame": (None, [vp, ci, ci, ci, ctypes.c_uint64, ci, ci]),
        "rd_synth_num_quads": (ci, [ci, ci]),
        # reference API (oclhelper.h / raw cl*)
        "simpleGetDevice": (vp, [ci]),
        "simpleCreateContext": (vp, [vp]),
        "allocatePinnedMemory": (vp, [cz, vp, vp]),
        "freePinnedMemory": (None, [vp, vp, vp]),
        "clCreateCommandQueue": (vp, [vp, vp, ctypes.c_ulong, vp]),
        "clReleaseCommandQueue": (ci, [vp]),
        "clReleaseContext": (ci, [vp]),
        "clCreateBuffer": (vp, [vp, ctypes.c_ulong, cz, vp, vp]),
        "clReleaseMemObject": (ci, [vp]),
        "clEnqueueReadBuffer": (ci, [vp, vp, ctypes.c_uint, cz, cz, vp, ctypes.c_uint, vp, vp]),
        "clEnqueueWriteBuffer": (ci, [vp, vp, ctypes.c_uint, cz, cz, vp, ctypes.c_uint, vp, vp]),
        "clFinish": (ci, [vp]),
        # oclimgutil.h
        "init_oclimgutil": (vp, [vp, vp]),
        "dispose_oclimgutil": (None, [vp]),
        "oclimgutil_clear": (vp, [vp, vp, ci, vp, vp]),
        "oclimgutil_copy": (vp, [vp, vp, vp, ci, vp, vp]),
        "oclimgutil_cast_i_f": (vp, [vp, vp, vp, cf, ci, vp, vp]),
        "oclimgutil_cast_c_i": (vp, [vp, vp, vp, ci, vp, vp]),
        "oclimgutil_threshold_i_i": (vp, [vp, vp, vp, ci, ci, ci, ci, vp, vp]),
        "oclimgutil_threshold_f_f": (vp, [vp, vp, vp, cf, cf, cf, ci, vp, vp]),
        "oclimgutil_convert_plab_bgr": (vp, [vp, vp, vp, ci, ci, ci, vp, vp]),
        "oclimgutil_unpack_f_f_f_plab": (vp, [vp, vp, vp, vp, vp, ci, ci, vp, vp]),
        "oclimgutil_pack_plab_f_f_f": (vp, [vp, vp, vp, vp, vp, ci, ci, vp, vp]),
        "oclimgutil_iirblur_f_f": (vp, [vp, vp, vp, vp, vp, ci, ci, ci, vp, vp]),
        "oclimgutil_edgevec_f2_f": (vp, [vp, vp, vp, ci, ci, vp, vp]),
        "oclimgutil_edge_f_plab": (vp, [vp, vp, vp, ci, ci, vp, vp]),
        "oclimgutil_thinthres_f_f_f2": (vp, [vp, vp, vp, vp, ci, ci, vp, vp]),
        "oclimgutil_label8x_int_int": (vp, [vp, vp, vp, vp, ci, ci, ci, vp, vp]),
        "oclimgutil_calcStrength": (vp, [vp, vp, vp, vp, ci, ci, vp, vp]),
        "oclimgutil_convert_bgr_lumaf": (vp, [vp, vp, vp, cf, ci, ci, ci, vp, vp]),
        "oclimgutil_convert_bgr_labeli": (vp, [vp, vp, vp, ci, ci, ci, ci, vp, vp]),
        "oclimgutil_convert_bgr_plab": (vp, [vp, vp, vp, ci, ci, ci, vp, vp]),
        "oclimgutil_edge_f_f": (vp, [vp, vp, vp, ci, ci, vp, vp]),
        "oclimgutil_edgevec_f2_plab": (vp, [vp, vp, vp, ci, ci, vp, vp]),
        "oclimgutil_thincubic_f_f_f2": (vp, [vp, vp, vp, vp, ci, ci, vp, vp]),
        "oclimgutil_filterStrength": (vp, [vp, vp, vp, ci, ci, ci, vp, vp]),
        # oclpolyline.h
        "init_oclpolyline": (vp, [vp, vp]),
        "dispose_oclpolyline": (None, [vp]),
        "oclpolyline_execute": (vp, [vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, cf, ci, ci, ci, vp, vp]),
        # oclrect.h
        "init_oclrect": (vp, [vp, vp, vp, vp, vp, ci, ci]),
        "dispose_oclrect": (None, [vp]),
        "oclrect_executeOnce": (vp, [vp, vp, ci, cd]),
        "oclrect_enqueueTask": (None, [vp, vp, ci]),
        "oclrect_pollTask": (vp, [vp, cd]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args


_libc = ctypes.CDLL(None)
_libc.free.argtypes = [ctypes.c_void_p]


def _take_rects(ptr):
    """Copy a malloc'd rect_t array (element 0 = header with nItems) into a numpy structured array and free it."""
    if not ptr:
        raise RuntimeError("detector returned NULL")
    n = ctypes.cast(ptr, ctypes.POINTER(ctypes.c_int))[0]
    buf = (ctypes.c_char * (176 * n)).from_address(ptr)
    out = np.frombuffer(buf, dtype=RECT_DTYPE).copy()[1:]
    _libc.free(ptr)
    return out


def gpu_available():
    return os.path.exists(LIB_PATH) and lib().rd_device_count() > 0


class Context:
    """device / context / queue triple created the way the reference programs do (rect.cpp:51-64)."""

    def __init__(self, did=0):
        L = lib()
        if L.rd_device_count() <= 0:
            raise RuntimeError("rectdetect_amd: no HIP device visible - there is no CPU fallback")
        self.device = L.simpleGetDevice(did)
        self.context = L.simpleCreateContext(self.device)
        self.queue = L.clCreateCommandQueue(self.context, self.device, 0, None)

    def buffer(self, array_or_bytes):
        L = lib()
        if isinstance(array_or_bytes, int):
            return L.clCreateBuffer(self.context, CL_MEM_READ_WRITE, array_or_bytes, None, None)
        a = np.ascontiguousarray(array_or_bytes)
        return L.clCreateBuffer(self.context, CL_MEM_READ_WRITE | CL_MEM_COPY_HOST_PTR, a.nbytes, a.ctypes.data, None)

    def read(self, mem, dtype, count):
        out = np.empty(count, dtype)
        rc = lib().clEnqueueReadBuffer(self.queue, mem, CL_TRUE, 0, out.nbytes, out.ctypes.data, 0, None, None)
        if rc != 0:
            raise RuntimeError("clEnqueueReadBuffer failed: %d" % rc)
        return out

    def pinned_copy(self, array):
        """a copy of `array` in page-locked host memory from the reference's own allocator (oclhelper.h: allocatePinnedMemory) as a numpy array; free with free_pinned()"""
        a = np.ascontiguousarray(array)
        p = lib().allocatePinnedMemory(a.nbytes, self.context, self.queue)
        out = np.ctypeslib.as_array((ctypes.c_uint8 * a.nbytes).from_address(p)).view(a.dtype).reshape(a.shape)
        out[...] = a
        return out

    def free_pinned(self, array):
        lib().freePinnedMemory(array.ctypes.data, self.context, self.queue)

    def release(self, *mems):
        for m in mems:
            lib().clReleaseMemObject(m)

    def close(self):
        lib().clReleaseCommandQueue(self.queue)
        lib().clReleaseContext(self.context)


def poly_frame(ctx, bgr, strength_thre=500, minerror=1.0, size_thre=20):
    """poly.cpp:68-131 on a BGR uint8 image (ih, iw, 3): returns (segments[LS_DTYPE] incl. header record, ids[ih*iw])."""
    L = lib()
    ih, iw = bgr.shape[:2]
    ws = bgr.strides[0]
    N = iw * ih
    img = np.zeros(N * 4, np.uint8)
    img[:ws * ih] = np.ascontiguousarray(bgr).reshape(-1)[:ws * ih]
    iu = L.init_oclimgutil(ctx.device, ctx.context)
    pl = L.init_oclpolyline(ctx.device, ctx.context)
    mem = [ctx.buffer(img)] + [ctx.buffer(N * 4) for _ in range(9)]
    memBig, memLS = ctx.buffer(N * 16), ctx.buffer(N * 16)
    q = ctx.queue
    L.oclimgutil_convert_plab_bgr(iu, mem[4], mem[0], iw, ih, ws, q, None)
    L.oclimgutil_unpack_f_f_f_plab(iu, mem[1], mem[2], mem[3], mem[4], iw, ih, q, None)
    L.oclimgutil_iirblur_f_f(iu, mem[0], mem[1], mem[4], mem[5], 2, iw, ih, q, None)
    L.oclimgutil_iirblur_f_f(iu, mem[1], mem[2], mem[4], mem[5], 2, iw, ih, q, None)
    L.oclimgutil_iirblur_f_f(iu, mem[2], mem[3], mem[4], mem[5], 2, iw, ih, q, None)
    L.oclimgutil_pack_plab_f_f_f(iu, mem[4], mem[0], mem[1], mem[2], iw, ih, q, None)
    L.oclimgutil_edgevec_f2_f(iu, memBig, mem[0], iw, ih, q, None)
    L.oclimgutil_edge_f_plab(iu, mem[5], mem[4], iw, ih, q, None)
    L.oclimgutil_thinthres_f_f_f2(iu, mem[2], mem[5], memBig, iw, ih, q, None)
    L.oclimgutil_threshold_f_f(iu, mem[9], mem[2], 0.0, 0.0, 1.0, N, q, None)
    L.oclimgutil_cast_i_f(iu, mem[8], mem[9], 1.0, N, q, None)
    L.oclimgutil_label8x_int_int(iu, mem[3], mem[8], mem[9], 0, iw, ih, q, None)
    L.oclimgutil_clear(iu, mem[4], N * 4, q, None)
    L.oclimgutil_calcStrength(iu, mem[4], mem[2], mem[3], iw, ih, q, None)
    L.oclimgutil_filterStrength(iu, mem[3], mem[4], strength_thre, iw, ih, q, None)
    L.oclimgutil_threshold_i_i(iu, mem[3], mem[3], 0, 0, 1, N, q, None)
    L.oclpolyline_execute(pl, memLS, N * 16, mem[0], mem[3], memBig, mem[4], mem[5], mem[6], mem[7], mem[8], mem[9],
                          minerror, size_thre, iw, ih, q, None)
    ids = ctx.read(mem[0], np.int32, N)
    hdr = ctx.read(memLS, np.int32, 14)
    n = int(hdr[0])
    segs = ctx.read(memLS, np.uint8, (n + 1) * 56).view(LS_DTYPE)
    ctx.release(memLS, memBig, *mem)
    L.dispose_oclpolyline(pl)
    L.dispose_oclimgutil(iu)
    return segs, ids


class RectDetector:
    """rect.cpp:78-105 / vidrect.cpp:128-172: the reference's oclrect API on host frames."""

    def __init__(self, ctx, iw, ih):
        L = lib()
        self.ctx, self.iw, self.ih = ctx, iw, ih
        self.iu = L.init_oclimgutil(ctx.device, ctx.context)
        self.pl = L.init_oclpolyline(ctx.device, ctx.context)
        self.h = L.init_oclrect(self.iu, self.pl, ctx.device, ctx.context, ctx.queue, iw, ih)

    def execute_once(self, bgr, tan_aov):
        a = np.ascontiguousarray(bgr)
        return _take_rects(lib().oclrect_executeOnce(self.h, a.ctypes.data, a.strides[0], float(tan_aov)))

    def enqueue(self, bgr):
        a = np.ascontiguousarray(bgr)
        self._keep = a
        lib().oclrect_enqueueTask(self.h, a.ctypes.data, a.strides[0])

    def poll(self, tan_aov):
        return _take_rects(lib().oclrect_pollTask(self.h, float(tan_aov)))

    def close(self):
        L = lib()
        L.dispose_oclrect(self.h)
        L.dispose_oclpolyline(self.pl)
        L.dispose_oclimgutil(self.iu)


class Detector:
    """The rd_detector extension: frames may already live in HBM, several frames in flight."""

    def __init__(self, iw, ih, device=0, nslots=2, nworkers=0, aperture=None):
        L = lib()
        if L.rd_device_count() <= 0:
            raise RuntimeError("rectdetect_amd: no HIP device visible - there is no CPU fallback")
        self.iw, self.ih, self.N = iw, ih, iw * ih
        self.h = L.rd_detector_create(device, iw, ih, nslots, nworkers)
        if aperture is not None:      # tan(AOV / 2) of the polls to come (rd_detector_set_aperture): work ahead of the first poll can use it
            L.rd_detector_set_aperture(self.h, float(aperture))

    def enqueue(self, frame, ws=None, on_device=False, pinned=False):
        """frame: a numpy BGR image (copied before the call returns) - or, with on_device / pinned, the ADDRESS of a frame in device memory / in pinned host memory
        (rd_host_alloc ...), which is read in place and must stay unchanged until the frame's poll returned"""
        if on_device or pinned:
            return lib().rd_detector_enqueue(self.h, frame, ws, 1 if on_device else 2)
        a = np.ascontiguousarray(frame)
        self._keep = a
        return lib().rd_detector_enqueue(self.h, a.ctypes.data, a.strides[0] if ws is None else ws, 0)

    def poll(self, tan_aov):
        return _take_rects(lib().rd_detector_poll(self.h, float(tan_aov)))

    def drain(self):
        lib().rd_detector_drain(self.h)

    def redone_frames(self):
        """frames whose polyline stage overflowed the single-launch kernel and was repeated the long way"""
        return lib().rd_detector_counter(self.h, 0)

    def frames_per_launch(self):
        """frames that share one set of launches (group launches, rd_detector_counter 15)"""
        return lib().rd_detector_counter(self.h, 15)

    def region_round_budget(self):
        """(current region-merge round budget, frames repeated with the full budget because theirs was too small)"""
        return lib().rd_detector_counter(self.h, 5), lib().rd_detector_counter(self.h, 4)

    def absorption(self):
        """(undecided pixels the tile kernel of the small-region absorption left to the single-block tail, sweeps the tail took) for the last
        polled frame, and how many frames so far were finished by the slow path (rd_detector_counter 14)"""
        w = self.plane("absorb", np.int32, 8)
        self.absorb_trace = {"steps": int(w[3]), "us_gather": w[4] / 100.0, "us_sweeps": w[5] / 100.0, "us_choose": w[6] / 100.0, "us_pointers": w[7] / 100.0}
        return int(w[1]), int(w[2]), lib().rd_detector_counter(self.h, 14)

    def device_time(self):
        """(summed device microseconds of the polled frames measured with HIP events, number of frames)"""
        return lib().rd_detector_counter(self.h, 1), lib().rd_detector_counter(self.h, 2)

    def last_segments(self):
        n = lib().rd_detector_last_segments(self.h, None, 0)
        if n < 0:
            return None
        out = np.zeros(n + 1, LS_DTYPE)
        lib().rd_detector_last_segments(self.h, out.ctypes.data, n + 1)
        return out

    def plane(self, name, dtype=np.int32, count=None):
        count = self.N if count is None else count
        out = np.zeros(count, dtype)
        got = lib().rd_detector_debug_plane(self.h, name.encode(), out.ctypes.data, out.nbytes)
        if got == 0:
            raise KeyError(name)
        return out

    def close(self):
        lib().rd_detector_destroy(self.h)


def postprocess_planes(segs, boundary, table, iw, ih, tan_aov):
    """Host post-process alone (oclrect.c:1049-1226 restated in csrc/rd_post.c) on full planes; runs without a GPU."""
    segs = np.ascontiguousarray(segs)
    boundary = np.ascontiguousarray(boundary, dtype=np.int32)
    table = np.ascontiguousarray(table, dtype=np.int32)
    return _take_rects(lib().rd_postprocess_planes(segs.ctypes.data, boundary.ctypes.data, table.ctypes.data, iw, ih, float(tan_aov)))
