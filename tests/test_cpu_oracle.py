"""CPU tests: the oracle (oracle/rd_oracle*.c, our restatement) against golden vectors that were produced by the
reference itself (tools/make_golden.py), the host post-process, the LUT closed forms and the synthetic generator."""
import glob
import os
import zlib

import numpy as np
import pytest

import rectdetect_amd as ra
from rectdetect_amd import synth
from tests import helpers


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def golden(name):
    return np.load(os.path.join(helpers.GOLDEN, name + ".npz"), allow_pickle=False)


RECT_FAST = ["rect_640x480_s0", "rect_640x480_s5", "rect_333x217_s2", "rect_1280x720_s1"]


@pytest.mark.parametrize("name", RECT_FAST)
def test_oracle_matches_reference_golden_planes(name):
    """Every intermediate plane of the restatement has the CRC the reference produced, frame after frame (H1 state)."""
    g = golden(name)
    iw, ih, N = int(g["iw"]), int(g["ih"]), int(g["iw"]) * int(g["ih"])
    orc = helpers.OracleRect(iw, ih)
    planes = [str(p) for p in g["planes"]]
    sizes = {"vxy": 2 * N, "table": (N * 4 // 5) * 5}
    for t in range(int(g["nframes"])):
        img = synth.frame(int(g["seed"]), iw, ih, t)
        assert crc(img) == int(g[f"f{t}_input_crc"])
        # parity is asserted where the reference's fixed-pass labelling had converged (last flag 0): SURVEY.md H4
        flags = g[f"f{t}_flags"]
        assert (flags[:, 10] == 0).all() and flags[3, 11] == 0
        orc.frame(img)
        got = [crc(orc.plane(p).view(np.uint32)[: sizes.get(p, N)]) for p in planes]
        bad = [p for p, a, b in zip(planes, got, g[f"f{t}_plane_crc"]) if a != int(b)]
        assert not bad, f"frame {t}: planes differ from the reference: {bad}"
        assert helpers.segments_equal(orc.segments(), g[f"f{t}_segments"])
    orc.close()


@pytest.mark.parametrize("name", RECT_FAST)
def test_postprocess_matches_reference_rectangles(name):
    """csrc/rd_post.c on the oracle's planes returns exactly the rect_t list the reference returned."""
    g = golden(name)
    iw, ih = int(g["iw"]), int(g["ih"])
    orc = helpers.OracleRect(iw, ih)
    for t in range(int(g["nframes"])):
        orc.frame(synth.frame(int(g["seed"]), iw, ih, t))
        rects = ra.postprocess_planes(orc.segments(), orc.plane("boundary"), orc.plane("table"), iw, ih, float(g["tan_aov"]))
        assert helpers.rects_equal(rects, g[f"f{t}_rects"])
    orc.close()


@pytest.mark.parametrize("name", ["poly_640x480_s0", "poly_333x217_s2", "poly_1280x720_s1_vid"])
def test_oracle_poly_path_matches_reference(name):
    g = golden(name)
    iw, ih = int(g["iw"]), int(g["ih"])
    img = synth.frame(int(g["seed"]), iw, ih, 0)
    assert crc(img) == int(g["input_crc"])
    segs, ids = helpers.oracle_poly(img, int(g["strength_thre"]), float(g["minerror"]), int(g["size_thre"]))
    assert helpers.segments_equal(segs, g["segments"])
    assert crc(ids) == int(g["ids_crc"])
    assert int(g["launches"]) == 163      # the reference's own launch count on this path


def test_reference_launch_count_recorded():
    assert int(golden("rect_640x480_s0")["f0_launches"]) == 220


def test_lut_closed_forms():
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_luts", os.path.join(helpers.ROOT, "tools", "gen_luts.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    O = helpers.oracle()
    import ctypes
    for which, tab in enumerate(m.tables()):
        n = ctypes.c_int()
        p = O.rdo_lut(which, ctypes.byref(n))
        assert np.array_equal(np.ctypeslib.as_array(p, (n.value,)), np.array(tab))
    # the generated header is up to date
    hdr = open(os.path.join(helpers.ROOT, "rectdetect_amd", "csrc", "rd_luts.h")).read()
    assert ",".join(str(v) for v in m.tables()[1][:16]) in hdr


def test_synth_c_and_numpy_twins_agree():
    L = ra.lib()
    for iw, ih, t in [(64, 48, 0), (333, 217, 5), (640, 480, 299)]:
        a = np.zeros((ih, iw * 3 + 5), np.uint8)
        L.rd_synth_frame(a.ctypes.data, iw, ih, iw * 3 + 5, synth.SEED0 + 3, t, 1)
        b = synth.frame(synth.SEED0 + 3, iw, ih, t)
        assert np.array_equal(a[:, : iw * 3].reshape(ih, iw, 3), b)
    assert synth.num_quads(1920, 1080) == 12 == L.rd_synth_num_quads(1920, 1080)
    assert synth.num_quads(640, 480) == 3


def test_oracle_edge_cases():
    """empty (flat) frame, tiny frame, frame narrower than a wavefront"""
    for iw, ih in [(64, 48), (40, 33), (130, 70)]:
        flat = np.full((ih, iw, 3), 40, np.uint8)
        orc = helpers.OracleRect(iw, ih)
        orc.frame(flat)
        assert int(orc.segments().view("i4")[0]) == 0
        rects = ra.postprocess_planes(orc.segments(), orc.plane("boundary"), orc.plane("table"), iw, ih, 0.7)
        assert len(rects) == 0
        orc.close()


def test_post_process_with_helper_threads_returns_the_same_bytes():
    """The pose estimations of a frame's candidates shared with helper threads (csrc/rd_post.c: armed by the poll that waits for the frame, spinning until the candidates
    are published) against the caller's thread alone: the same rect_t bytes in the same order, whoever computed which candidate - armed and not armed (helpers asleep: the
    caller runs everything), and from two threads at once (one finds the helpers taken and runs alone)."""
    import threading
    L = ra.lib()
    iw, ih = 640, 480
    cases = []
    for seed in (0, 5):
        orc = helpers.OracleRect(iw, ih)
        orc.frame(synth.frame(synth.SEED0 + seed, iw, ih, 0))
        cases.append((orc.segments().copy(), orc.plane("boundary").copy(), orc.plane("table").copy()))
        orc.close()
    run = lambda c: ra.postprocess_planes(c[0], c[1], c[2], iw, ih, 0.7).tobytes()
    alone = [run(c) for c in cases]      # (no helper exists yet in this process, unless an earlier test made some: then this is "not armed")
    assert any(len(a) >= 2 * 176 for a in alone), "the frames should have several candidates"
    L.rd_post_helpers_configure(3)
    assert L.rd_post_helpers() >= 3
    for rep in range(20):
        for c, want in zip(cases, alone):
            if rep % 2 == 0:
                L.rd_post_helpers_arm()
            assert run(c) == want
    bad = []
    def worker(i):
        for rep in range(20):
            L.rd_post_helpers_arm()
            if run(cases[i]) != alone[i]:
                bad.append(i)
    ths = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not bad


@pytest.mark.ref
def test_oracle_against_live_reference_new_seed():
    """where the reference build exists: a seed that has no golden file, compared plane by plane"""
    iw, ih = 320, 240
    img = synth.frame(synth.SEED0 + 11, iw, ih, 3)
    r = helpers.RefRect(iw, ih)
    names = ["plab0", "Lblur", "plab1", "vxy", "strength", "nms", "tidy", "str_sum", "smooth", "quant", "strong", "mergemask", "region", "boundary", "lsid", "table"]
    rects, snaps = r.execute_once(img, 0.7, snapshots=names + ["lslist"])
    r.close()
    orc = helpers.OracleRect(iw, ih)
    orc.frame(img)
    N = iw * ih
    sizes = {"vxy": 2 * N, "table": (N * 4 // 5) * 5}
    for p in names:
        k = sizes.get(p, N)
        assert np.array_equal(orc.plane(p).view(np.uint32)[:k], snaps[p][:k]), p
    mine = ra.postprocess_planes(orc.segments(), orc.plane("boundary"), orc.plane("table"), iw, ih, 0.7)
    assert helpers.rects_equal(mine, rects)
    orc.close()


@pytest.mark.parametrize("name", ["ops_97x61", "ops_160x131"])
def test_oracle_operators_against_reference_golden(name):
    """every oclimgutil.h operator the oracle restates, on the inputs stored with the reference's outputs
    (tools/make_golden_ops.py ran the reference's own kernels): bit-exact, floats included"""
    g = golden(name)
    iw, ih = int(g["iw"]), int(g["ih"])
    N, ws = iw * ih, iw * 3 + 1
    O = helpers.oracle()
    P = helpers.P
    f, plab, lab = (np.ascontiguousarray(g[k]) for k in ("in_f", "in_plab", "in_lab"))
    bits = lambda a: np.ascontiguousarray(a).view(np.uint32) if a.dtype == np.float32 else a

    def bgr_out(fn, *args):
        out = np.zeros((ih, ws), np.uint8)
        fn(P(out), *args, iw, ih, ws)
        return out[:, : iw * 3]

    import ctypes
    O.rdo_convert_bgr_lumaf.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    assert np.array_equal(bgr_out(O.rdo_convert_bgr_lumaf, P(f), 0.9), g["convert_bgr_lumaf"])
    assert np.array_equal(bgr_out(O.rdo_convert_bgr_labeli, P(lab), -1), g["convert_bgr_labeli"])
    assert np.array_equal(bgr_out(O.rdo_plab2bgr, P(plab)), g["convert_bgr_plab"])
    out = np.zeros(N, np.float32)
    O.rdo_edge_f_f(P(out), P(f), iw, ih)
    assert np.array_equal(bits(out), bits(g["edge_f_f"]))
    v = np.zeros(2 * N, np.float32)
    O.rdo_edgevec_plab(P(v), P(plab), iw, ih)
    assert np.array_equal(bits(v), bits(g["edgevec_f2_plab"]))
    vf = np.zeros(2 * N, np.float32)
    O.rdo_edgevec(P(vf), P(f), iw, ih)
    assert np.array_equal(bits(vf), bits(g["edgevec_f2_f"]))
    O.rdo_thincubic(P(out), P(f), P(vf), iw, ih)
    assert np.array_equal(bits(out), bits(g["thincubic_f_f_f2"]))
    O.rdo_thinthres(P(out), P(f), P(vf), iw, ih)
    assert np.array_equal(bits(out), bits(g["thinthres_f_f_f2"]))
    O.rdo_edge_plab(P(out), P(plab), iw, ih)
    assert np.array_equal(bits(out), bits(g["edge_f_plab"]))
    O.rdo_iirblur(P(out), P(f), iw, ih)
    assert np.array_equal(bits(out), bits(g["iirblur_f_f"]))
    pl = np.zeros(N, np.uint32)
    O.rdo_bgr2plab(P(pl), P(np.ascontiguousarray(g["in_bgr"])), iw, ih, iw * 3)
    assert np.array_equal(pl, g["convert_plab_bgr"])
    u = [np.zeros(N, np.float32) for _ in range(3)]
    O.rdo_unpack_plab(P(u[0]), P(u[1]), P(u[2]), P(plab), N)
    for k in range(3):
        assert np.array_equal(bits(u[k]), bits(g["unpack%d" % k]))
    O.rdo_pack_plab(P(pl), P(u[0]), P(u[1]), P(u[2]), N)
    assert np.array_equal(pl, g["pack_plab_f_f_f"])
    acc = (np.arange(N, dtype=np.int32) % 3).astype(np.int32)
    O.rdo_calc_strength(P(acc), P(f), P(lab), iw, ih)
    assert np.array_equal(acc, g["calcStrength"])
    l2 = lab.copy()
    O.rdo_filter_strength(P(l2), P(acc), 5000, iw, ih)
    assert np.array_equal(l2.ravel(), g["filterStrength"])


@pytest.mark.parametrize("name", ["ops_iir_97x61", "ops_iir_160x131"])
def test_oracle_iir_blur_every_radius_against_reference_golden(name):
    """oclimgutil_iirblur_f_f with radii no application passes (sigma = (r + 1) / 3, r = 0 .. 31): the reference's own
    kernels made the expected planes (tools/make_golden_ops.py); bit-exact"""
    g = golden(name)
    iw, ih = int(g["iw"]), int(g["ih"])
    O, P = helpers.oracle(), helpers.P
    f = np.ascontiguousarray(g["in_f"])
    for r in [int(r) for r in g["radii"]]:
        out = np.zeros(iw * ih, np.float32)
        assert O.rdo_iirblur_r(P(out), P(f), iw, ih, r) == 0
        assert np.array_equal(out.view(np.uint32), np.ascontiguousarray(g["r%d" % r]).view(np.uint32)), r
    # r = 2 through the table == the sigma = 1 entry point every caller uses
    a, b = np.zeros(iw * ih, np.float32), np.zeros(iw * ih, np.float32)
    O.rdo_iirblur(P(a), P(f), iw, ih)
    O.rdo_iirblur_r(P(b), P(f), iw, ih, 2)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # outside the domain the reference defines: unknown radius, lines shorter than the mirrored warm-up
    for r, w, h in [(-1, iw, ih), (32, iw, ih), (5, 15, ih), (5, iw, 15)]:
        assert O.rdo_iirblur_r(P(out), P(f), w, h, r) == -1


def test_iir_table_row2_is_the_fast_paths():
    """the generated table (tools/gen_iircoef.py) and the constants the sigma = 1 path was written with agree bit for bit"""
    O = helpers.oracle()
    row2 = np.array([O.rdo_iircoef(2)[k] for k in range(15)], np.float32)
    own = np.array([O.rdo_iircoef(-1)[k] for k in range(15)], np.float32)
    assert np.array_equal(row2.view(np.uint32), own.view(np.uint32))
    # and the product's copy of the table is the oracle's (same generator, two outputs)
    a = open(os.path.join(helpers.ROOT, "oracle", "rd_iircoef.h")).read()
    b = open(os.path.join(helpers.ROOT, "rectdetect_amd", "csrc", "rd_iircoef.h")).read()
    assert a == b
    import re
    src = open(os.path.join(helpers.ROOT, "rectdetect_amd", "csrc", "rd_k_front.hip")).read()
    fast = [np.float32(re.search(r"#define IIR_C%d (-?[0-9.]+)f" % k, src).group(1)) for k in range(15)]
    assert np.array_equal(np.array(fast, np.float32).view(np.uint32), own.view(np.uint32))


def test_region_spec_against_the_references_own_order_dependence():
    """The order-free region schedule (oracle REGION_SPEC mode = what the HIP path implements) against the reference's own
    rectangle lists under 26 legal work-item orders of its two in-place region kernels (tests/golden/hard_rect_orders.npz,
    order 0 = serial raster): on a busy frame the reference returns 32..36 rectangles depending on the order; the spec must
    contain every rectangle common to all orders and nothing that no order produces.  (All 14 frames: GPU test
    test_busy_inputs_final_outputs_vs_reference; here the first tile frame and one order-independent frame, to bound the time.)"""
    g, go = golden("hard_rect"), golden("hard_rect_orders")
    key = lambda r: r["c2"].tobytes() + r["c3"].tobytes() + r["value"].tobytes() + r["status"].tobytes()
    tan = float(np.tan(36.0 / 180.0 * np.pi))
    for hi in (0, 9):
        kind = g["kinds"].tolist()[hi]
        seed, iw, ih = g["params"].tolist()[hi]
        union, member = go["h%d_union" % hi], go["h%d_member" % hi]
        ukeys = [key(r) for r in union]
        assert set(k for k, m in zip(ukeys, member[0]) if m) == set(key(r) for r in g["h%d_rects" % hi])      # order 0 is the raster fixture
        orc = helpers.OracleRect(iw, ih, helpers.REGION_SPEC)
        orc.frame(synth.hard_frame(kind, seed, iw, ih))
        assert helpers.segments_equal(orc.segments(), g["h%d_segments" % hi])
        assert orc.rounds()[0] < 20
        rects = ra.postprocess_planes(orc.segments(), orc.plane("boundary"), orc.plane("table"), iw, ih, tan)
        here = set(key(r) for r in rects)
        stable = set(k for k, m in zip(ukeys, member.all(0)) if m)
        assert stable <= here <= set(ukeys)
        if hi == 0:
            assert member.sum(1).min() < member.sum(1).max() and len(stable) < len(here)     # the reference does depend on the order here
        orc.close()


@pytest.mark.parametrize("iw,ih,seed", [(333, 217, 2), (640, 480, 5)])
def test_region_spec_stage_relations(iw, ih, seed):
    """REGION_SPEC mode: the merge kernel with concurrent work-items settles well inside the 20 launches the HIP path budgets for; its
    first 8 launches are the REGION_REFERENCE_CONCURRENT plane; the fixed point of the Jacobi absorption IS the serial raster result of
    the reference's kernel (same inputs) - the argument by which the HIP path's parallel evaluation of it is exact; the planes
    downstream follow by the order-independent stages"""
    O, P = helpers.oracle(), helpers.P
    N = iw * ih
    img = synth.frame(synth.SEED0 + seed, iw, ih, 0)
    orc = helpers.OracleRect(iw, ih, helpers.REGION_SPEC)
    orc.frame(img)
    launches, absorb_rounds = orc.rounds()
    assert 2 <= launches < 20 and absorb_rounds == 0
    quant, mask, edge, junction = (orc.plane(n).view(np.int32) for n in ("quant", "mergemask", "label1", "junction"))
    lab = np.zeros(N, np.int32)
    assert O.rdo_region_concurrent(P(lab), P(quant), P(mask), P(edge), iw, ih, 1000) == launches        # 64 was not the limit
    one_less = np.zeros(N, np.int32)
    O.rdo_region_concurrent(P(one_less), P(quant), P(mask), P(edge), iw, ih, launches - 1)
    assert np.array_equal(one_less, lab), "the last launch must have changed nothing"
    inner = np.zeros((ih, iw), bool)
    inner[1:-1, 1:-1] = True
    assert np.array_equal(lab[lab][inner.reshape(-1)], lab[inner.reshape(-1)]), "at the fixed point every processed pixel points at a root"
    size = junction.copy()
    O.rdo_region_size(P(size), P(lab), N)
    assert np.array_equal(size, orc.plane("rsize"))
    serial, jac = lab.copy(), lab.copy()
    O.rdo_despeckle2(P(serial), P(size), 16, iw, ih)
    rounds = O.rdo_despeckle2_jacobi_k(P(jac), P(size), 16, iw, ih, None, 1 << 30)
    assert np.array_equal(serial, jac), "the fixed point of the Jacobi rounds must be the raster-order result"
    assert rounds >= 1 and np.array_equal(serial, orc.plane("region"))      # the spec's absorption IS the serial raster result
    marks = np.zeros(N, np.int32)
    O.rdo_mark_boundary(P(marks), P(orc.plane("region")), iw, ih)
    assert np.array_equal(marks, orc.plane("boundary_src"))
    orc.close()


@pytest.mark.ref
@pytest.mark.parametrize("iw,ih,seed,nframes", [(333, 217, 2, 2), (640, 480, 0, 2)])
def test_concurrent_merge_restatement_equals_the_references_own_kernel(iw, ih, seed, nframes):
    """pins rdo_region_concurrent: the reference itself (oracle/_ref: its kernels and host code) with the work-items of labelMergeMain
    running concurrently (rdcl_set_order group order 5: every work-item reads the labels the launch began with, the atomic minima take
    effect together) against the oracle in REGION_REFERENCE_CONCURRENT mode - region, size and boundary planes and the rectangle list,
    bit for bit, on consecutive frames (the state carried from frame to frame included)"""
    import ctypes
    R = helpers.ref()
    R.rdcl_set_order.argtypes = [ctypes.c_char_p] + [ctypes.c_int] * 4
    tan = float(np.tan(36.0 / 180.0 * np.pi))
    r = helpers.RefRect(iw, ih)
    orc = helpers.OracleRect(iw, ih, helpers.REGION_REFERENCE_CONCURRENT)
    try:
        R.rdcl_set_order(b"rect:labelMergeMain", 0, 0, 5, 0)
        for t in range(nframes):
            img = synth.frame(synth.SEED0 + seed, iw, ih, t)
            rects, snaps = r.execute_once(img, tan, snapshots=["region", "rsize", "boundary", "table"])
            orc.frame(img)
            for k in ("region", "rsize", "boundary"):
                assert np.array_equal(snaps[k].view(np.int32), orc.plane(k).view(np.int32)), (t, k)
            mine = ra.postprocess_planes(orc.segments(), orc.plane("boundary"), orc.plane("table"), iw, ih, tan)
            assert helpers.rects_equal(rects, mine), t
    finally:
        R.rdcl_set_order(b"", 0, 0, 0, 0)
        r.close()
        orc.close()


@pytest.mark.ref
@pytest.mark.parametrize("iw,ih,seed,nframes", [(333, 217, 2, 2), (640, 480, 5, 2), (640, 480, 0, 1)])
def test_host_postprocess_equals_the_references_own_compiled_host_code(iw, ih, seed, nframes):
    """pins csrc/rd_post.c (+ rd_post_core.h) from the outside: THE REFERENCE'S OWN compiled executeCPUTask (oclrect.c:1049-1226, unchanged, oracle/_ref) runs on the very
    planes rd_post.c is given - the stand-in hands them to the three read-backs of genGPUTask (oclrect.c:371-376) and runs none of the 220 launches
    (helpers.RefRect.host_postprocess) - and the two lists must agree in every bit AND in order; on the planes of the raster-order reference and on those of the concurrent,
    settled region merge (what the HIP path computes), for two apertures.  The GPU suite repeats this on the HIP path's own planes over the long streams
    (test_references_own_host_code_on_the_hip_paths_planes)."""
    r = helpers.RefRect(iw, ih)
    checked = 0
    for mode in (helpers.REGION_REFERENCE_RASTER, helpers.REGION_SPEC):
        orc = helpers.OracleRect(iw, ih, mode)
        for t in range(nframes):
            orc.frame(synth.frame(synth.SEED0 + seed, iw, ih, t))
            segs, boundary, table = orc.segments(), orc.plane("boundary"), orc.plane("table")
            for half_aov in (36.0, 25.0):
                tan = float(np.tan(half_aov / 180.0 * np.pi))
                mine = ra.postprocess_planes(segs, boundary, table, iw, ih, tan)
                theirs = r.host_postprocess(segs, boundary, table, tan)
                assert helpers.rects_equal(mine, theirs), (mode, t, half_aov, len(mine), len(theirs))
                checked += len(mine)
        orc.close()
    r.close()
    assert checked > 0


def test_builtin_sensitivity_fixture_is_consistent_with_the_other_goldens():
    """tests/golden/builtin_sensitivity.npz (tools/make_golden_builtins.py: the reference under other legal OpenCL builtin choices): variant 0 is the
    baseline every other fixture was made with - its rectangle lists must be the ones of the still fixtures of the same frames - and the recorded
    differences of the baseline against itself are zero."""
    g = golden("builtin_sensitivity")
    variants = [str(v) for v in g["variants"]]
    assert variants[0] == "baseline" and len(variants) >= 5
    key = lambda r: r["c2"].tobytes() + r["c3"].tobytes() + r["value"].tobytes() + r["status"].tobytes()
    stills = {(0, 640, 480, 0): "rect_640x480_s0", (5, 640, 480, 0): "rect_640x480_s5", (1, 1280, 720, 0): "rect_1280x720_s1", (0, 1920, 1080, 0): "rect_1920x1080_s0"}
    checked = 0
    for fi, fr in enumerate(g["frames"].tolist()):
        member, union = g[f"f{fi}_member"], g[f"f{fi}_union"]
        assert member.shape == (len(variants), len(union)) and not g[f"f{fi}_planes"][0].any() and g[f"f{fi}_segs"][0, 1] == 0
        name = stills.get(tuple(fr))
        if name and float(golden(name)["tan_aov"]) == float(np.tan(36.0 / 180.0 * np.pi)):
            ref = golden(name)["f0_rects"]
            assert set(key(r) for r, m in zip(union, member[0]) if m) == set(key(r) for r in ref), name      # (as sets: a list may hold the same rectangle twice)
            checked += 1
    assert checked >= 2
