"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU and exports every function that
include/*.h declares; calling into the detector without a GPU fails loudly (no CPU fallback)."""
import ctypes
import glob
import os
import re
import subprocess
import sys

import pytest

import rectdetect_amd as ra
from tests import helpers

HDRS = sorted(glob.glob(os.path.join(helpers.ROOT, "include", "*.h")))


def declared_functions(path):
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    txt = re.sub(r"#[^\n]*", "", txt)
    names = set()
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_]*)\s*\(([^;{}()]|\([^()]*\))*\)\s*;", txt):
        n = m.group(1)
        if n not in ("sizeof", "defined", "for", "if", "return", "while") and not n.startswith("RD_"):
            names.add(n)
    return names


def test_headers_present():
    base = {os.path.basename(h) for h in HDRS}
    assert {"helper.h", "oclhelper.h", "oclimgutil.h", "oclpolyline.h", "oclrect.h", "vec234.h", "rectdetect_hip.h"} <= base


@pytest.mark.parametrize("hdr", [h for h in HDRS if not h.endswith("vec234.h")])
def test_every_declared_symbol_is_exported(hdr):
    L = ctypes.CDLL(ra.LIB_PATH)
    names = declared_functions(hdr)
    assert len(names) >= 3, (hdr, names)
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, f"{os.path.basename(hdr)} declares symbols the library does not export: {missing}"


def test_reference_surface_counts():
    """24 operator wrappers + init/dispose in oclimgutil.h (reference oclimgutil.h:74-100)"""
    names = declared_functions(os.path.join(helpers.ROOT, "include", "oclimgutil.h"))
    assert len([n for n in names if n.startswith("oclimgutil_")]) == 24
    assert {"init_oclimgutil", "dispose_oclimgutil"} <= names


def test_headers_compile_as_c_and_cpp(tmp_path):
    src = '#define CL_TARGET_OPENCL_VERSION 120\n#include <stdint.h>\n#include <CL/cl.h>\n#include "vec234.h"\n#include "helper.h"\n#include "oclhelper.h"\n' \
          '#include "oclimgutil.h"\n#include "oclpolyline.h"\n#include "oclrect.h"\n#include "rectdetect_hip.h"\n' \
          'int main(void){ rect_t r; linesegment_t l; (void)r; (void)l; return sizeof(rect_t) == 176 && sizeof(linesegment_t) == 56 ? 0 : 1; }\n'
    for comp, ext in (("gcc", "c"), ("g++", "cpp")):
        f = tmp_path / f"t.{ext}"
        f.write_text(src)
        exe = tmp_path / f"t_{ext}"
        subprocess.check_call([comp, "-I", os.path.join(helpers.ROOT, "include"), str(f), "-o", str(exe)])
        assert subprocess.call([str(exe)]) == 0


def test_no_gpu_means_loud_failure():
    """In a process without a HIP device the detector refuses to start instead of computing on the CPU."""
    code = "import rectdetect_amd as ra, sys\n" \
           "sys.exit(3) if ra.lib().rd_device_count() > 0 else None\n" \
           "ra.Detector(64, 48)\n"
    p = subprocess.run([sys.executable, "-c", code], cwd=helpers.ROOT, capture_output=True, text=True)
    if p.returncode == 3:
        pytest.skip("a GPU is visible here")
    assert p.returncode != 0
    assert "no CPU" in (p.stderr + p.stdout) or "no HIP device" in (p.stderr + p.stdout)


def test_product_does_not_reference_the_oracle():
    """nothing under rectdetect_amd/ or include/ may import, link or load anything from oracle/"""
    bad = []
    for root in ("rectdetect_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(helpers.ROOT, root)):
            if "build" in dp:
                continue
            for fn in fns:
                if fn.endswith((".so", ".o", ".pyc")):
                    continue
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                if re.search(r"rd_oracle|librd_oracle|librdref|oracle/", txt):
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad
    out = subprocess.check_output(["ldd", ra.LIB_PATH], text=True)
    assert "oracle" not in out and "rdref" not in out


def test_example_programs_build_against_the_headers():
    """examples/rdrect.c and rdvid.c are written against include/*.h only (the reference's API) and link with the library"""
    subprocess.check_call(["make", "-C", os.path.join(helpers.ROOT, "examples")], stdout=subprocess.DEVNULL)
    for exe in ("rdrect", "rdvid", "rdpoly"):
        assert os.access(os.path.join(helpers.ROOT, "examples", exe), os.X_OK)


def test_blur_division_formula_is_exact():
    """the edge-stopped blur replaces floor(s / w) by trunc(fma(float(s), 1/w, 0.5/w)) (rd_k_rect.hip: div_small_f); the formula is
    checked for every operand pair the kernel can produce"""
    out = subprocess.run([sys.executable, os.path.join(helpers.ROOT, "tools", "check_div_small.py")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "exact" in out.stdout, out.stdout + out.stderr


REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.exists(os.path.join(REFERENCE, "rect.cpp")), reason="reference sources not present (only in the build container)")
@pytest.mark.parametrize("app", ["rect", "poly", "vidrect", "vidpoly"])
def test_reference_applications_compile_and_link_unchanged(app, tmp_path):
    """north_star: "rect.cpp, poly.cpp and vidrect.cpp link unchanged".  The reference's demo programs are compiled WHERE THEY LIE
    (nothing of them is copied) against include/*.h and linked with librectdetect_hip.so instead of libOpenCL + the reference's own
    objects.  OpenCV is absent from this image, so its handful of names comes from a declarations-only stand-in
    (tests/opencv_stub/opencv2/opencv.hpp); everything else - every oclrect / oclpolyline / oclimgutil / oclhelper / helper / cl*
    symbol the programs import - must be resolved by the library.  The program then starts and prints its usage."""
    obj, exe = tmp_path / (app + ".o"), tmp_path / app
    subprocess.check_call(["g++", "-std=gnu++11", "-DCL_TARGET_OPENCL_VERSION=120", "-w", "-I", os.path.join(helpers.ROOT, "tests", "opencv_stub"),
                           "-I", os.path.join(helpers.ROOT, "include"), "-c", os.path.join(REFERENCE, app + ".cpp"), "-o", str(obj)])
    # what the program imports beyond libc / libstdc++: all of it must come from the library
    L = ctypes.CDLL(ra.LIB_PATH)
    undefined = [l.split()[-1] for l in subprocess.check_output(["nm", "-u", str(obj)], text=True).splitlines()]
    ours = [u for u in undefined if re.match(r"(cl[A-Z]|ocl|init_ocl|dispose_ocl|simple|exitf|ce$|getDeviceName|loadPlan|savePlan|startProfiling|finishProfiling|"
                                             r"currentTimeMillis|allocatePinnedMemory|freePinnedMemory|waitForEvent|clStrError|checkError)", u)]
    assert len(ours) >= 8, ours
    missing = [u for u in ours if not hasattr(L, u)]
    assert not missing, f"{app}.cpp imports symbols the library does not export: {missing}"
    subprocess.check_call(["g++", str(obj), "-o", str(exe), "-L", os.path.dirname(ra.LIB_PATH), "-lrectdetect_hip", "-Wl,-rpath," + os.path.dirname(ra.LIB_PATH), "-lm"])
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert "Usage" in (p.stdout + p.stderr) or "usage" in (p.stdout + p.stderr), (p.returncode, p.stdout[-500:], p.stderr[-500:])


def test_probe_geometry_against_the_oracles_independent_restatement():
    """oclrect.c:1066-1083: the 15 probe pixels per segment.  The product's sampling kernel and its test tap share ONE helper (rd_post_core.h:
    rdp_probe_pixel); the oracle restates the reference's loop on its own (rd_oracle.c: rdo_probe_pixels).  Random segments, segments on and across
    the frame's border, degenerate ones: every pixel must agree."""
    import ctypes
    import numpy as np
    L = ra.lib()
    O = helpers.oracle()
    f, i, p = ctypes.c_float, ctypes.c_int, ctypes.c_void_p
    L.rd_probe_pixels.argtypes = [f, f, f, f, i, i, p]
    L.rd_probe_pixels.restype = None
    O.rdo_probe_pixels.argtypes = [f, f, f, f, i, i, p]
    O.rdo_probe_pixels.restype = None
    rng = np.random.default_rng(11)
    iw, ih = 1920, 1080
    cases = [(0.0, 0.0, 0.0, 0.0), (5.5, 5.5, 5.5, 5.5), (0.4, 0.4, 1919.6, 1079.6), (-3.0, 10.0, 30.0, -2.0), (1918.5, 0.0, 1919.49, 1079.0), (100.5, 200.5, 100.5, 900.5), (2.5, 3.5, 4.5, 3.5)]
    cases += [tuple(float(v) for v in (rng.uniform(-4, iw + 4), rng.uniform(-4, ih + 4), rng.uniform(-4, iw + 4), rng.uniform(-4, ih + 4))) for _ in range(20000)]
    cases += [tuple(float(np.float32(v)) for v in (x, y, x + rng.uniform(-3, 3), y + rng.uniform(-3, 3))) for x, y in zip(rng.uniform(0, iw, 5000), rng.uniform(0, ih, 5000))]
    a, b = np.zeros(30, np.int32), np.zeros(30, np.int32)
    for c in cases:
        L.rd_probe_pixels(*c, iw, ih, a.ctypes.data)
        O.rdo_probe_pixels(*c, iw, ih, b.ctypes.data)
        assert np.array_equal(a, b), (c, a.tolist(), b.tolist())
