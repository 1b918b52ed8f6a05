"""GPU parity tests (-m gpu): the HIP path, driven through the C ABI, against the oracle on the same seeded inputs and
against golden vectors produced by the reference.  Integer / byte / index planes and the float planes are compared
BIT-EXACTLY (the kernels use the oracle's operation order, no FMA contraction, correctly rounded sqrt and divide)."""
import os
import zlib

import numpy as np
import pytest

import rectdetect_amd as ra
from rectdetect_amd import synth
from tests import helpers

pytestmark = pytest.mark.gpu

# planes that must be bit-identical to the oracle in its reference mode (= the reference's kernels, work-items in raster order)
EXACT = [("plab0", "plab0", 1), ("lblur", "Lblur", 1), ("plab1", "plab1", 1), ("vxy", "vxy", 2), ("strength", "strength", 1), ("nms", "nms", 1),
         ("mask0", "mask0", 1), ("tidy", "tidy", 1), ("strsum", "str_sum", 1), ("edge500", "edge500", 1), ("smooth", "smooth", 1), ("quant", "quant", 1),
         ("strong", "strong", 1), ("label1", "label1", 1), ("junction", "junction", 1), ("mergemask", "mergemask", 1), ("lsid", "lsid", 1)]
# The region stages: the reference's labelMergeMain / despeckle2 update labels in place, so their result depends on the order
# in which a device runs the work-items (tests/golden/hard_rect_orders.npz shows the reference's own
# rectangle lists changing with it).  The HIP path reproduces ONE legal execution of the reference's kernels bit for bit: the
# merge kernel with the work-items of a launch running concurrently, launched until a launch changes nothing, then the
# absorption exactly as the serial raster order gives it.  Normative definition: the oracle's REGION_SPEC mode
# (oracle/rd_oracle.c: rdo_region_concurrent - itself pinned to the reference's own kernel running that way, tests/test_cpu_oracle.py -
# and rdo_despeckle2): these planes must be bit-identical to THAT.
REGION_EXACT = [("region0", None), ("region", "region"), ("rsize", "rsize"), ("boundarysrc", "boundary_src"), ("boundary", "boundary"), ("table", "table")]
TAN36 = float(np.tan(36.0 / 180.0 * np.pi))


def detector(iw, ih, device_post=False, tan=None, **kw):
    """a detector; device_post: the rectangles come from the device (RD_DEVICE_POST=1: candidate funnel + pose estimation in
    rd_k_post.hip) - the aperture of the polls to come is announced so that the first frame takes that path as well"""
    if not device_post:
        return ra.Detector(iw, ih, **kw)
    old = os.environ.get("RD_DEVICE_POST")
    os.environ["RD_DEVICE_POST"] = "1"
    try:
        return ra.Detector(iw, ih, aperture=TAN36 if tan is None else tan, **kw)
    finally:
        os.environ.pop("RD_DEVICE_POST", None) if old is None else os.environ.__setitem__("RD_DEVICE_POST", old)


# frames of the long streams whose rectangle SET is bit-identical to the raster-order reference's (recorded; must not drop)
EXACT_FRAMES_MIN = {"stream_1280x720_s1_300": 298, "stream_1920x1080_s0_100": 96, "stream_3840x2160_s4_16": 16, "stream_1920x1080_s7_100": 94, "stream_1920x1080_s0": 16, "stream_1280x720_s1": 30, "stream_3840x2160_s4": 3,
                    "stream_1920x1080_s11_200": 191, "stream_1920x1080_s12_200": 195, "stream_1280x720_s13_300": 294,
                    "stream_1920x1080_s21_300": 287, "stream_1920x1080_s22_300": 271,
                    "stream_1920x1080_s23_300": 264, "stream_1920x1080_s24_300": 277, "stream_3840x2160_s4_100": 69}
# the second golden of every long stream (tests/golden/<stream>_settled.npz, tools/make_golden_settled.py): what THE REFERENCE'S OWN compiled code returns when its merge
# kernel runs with concurrent work-items (order 26 of stream_orders.npz) and is launched until it settles - the execution the HIP path reproduces.  The short streams are
# the first frames of the long ones.
SETTLED_GOLDEN = {"stream_1920x1080_s0": "stream_1920x1080_s0_100", "stream_1280x720_s1": "stream_1280x720_s1_300", "stream_3840x2160_s4": "stream_3840x2160_s4_16"}
rect_key = lambda r: r["c2"].tobytes() + r["c3"].tobytes() + r["value"].tobytes() + r["status"].tobytes()


def settled_golden(name):
    return golden(SETTLED_GOLDEN.get(name, name) + "_settled")


def inside_order_dependence(rects, go, name, t):
    """the reference's own lists of this frame under the 32 sampled work-item orders (stream_orders.npz, as multisets: `_count[order, i]` = how often the list of that order
    holds rectangle i): every rectangle here is one the reference returns under some order, as often as under SOME order at least and at most; (inside?, reference independent?)"""
    if f"{name}_f{t}_union" not in go.files:
        return None, None
    union, count = go[f"{name}_f{t}_union"], go[f"{name}_f{t}_count"].astype(np.int64)
    index = {rect_key(r): i for i, r in enumerate(union)}
    here = np.zeros(len(union), np.int64)
    for r in rects:
        if rect_key(r) not in index:
            return False, bool((count == count[0]).all())
        here[index[rect_key(r)]] += 1
    return bool((here >= count.min(0)).all() and (here <= count.max(0)).all()), bool((count == count[0]).all())


def check_region_planes(det, orc, where=""):
    """the five region-stage planes of the detector's last frame against the spec-mode oracle `orc` (same frame), plus
    stage-isolated checks that feed each oracle stage with the GPU's own input plane (they hold whatever the merge did)"""
    N, iw, ih = det.iw * det.ih, det.iw, det.ih
    O, P = helpers.oracle(), helpers.P
    nentry = N * 4 // 5
    gp = {g: det.plane(g, np.int32, nentry * 5 if g == "table" else N) for g, _ in REGION_EXACT}
    gp["junction"], gp["lsid"] = det.plane("junction"), det.plane("lsid")
    for g, o in REGION_EXACT:
        if o is None:
            continue
        b = orc.plane(o).view(np.int32)[: len(gp[g])]
        assert np.array_equal(gp[g], b), f"{where}: plane {g} differs from the spec in {int((gp[g] != b).sum())} elements"
    # stage by stage on the GPU's own planes
    size = gp["junction"].copy()
    O.rdo_region_size(P(size), P(gp["region0"]), N)
    assert np.array_equal(size, gp["rsize"]), f"{where}: rsize != junction + histogram(region0)"
    lab = gp["region0"].copy()
    O.rdo_despeckle2(P(lab), P(gp["rsize"]), 16, iw, ih)
    assert np.array_equal(lab, gp["region"]), f"{where}: region != the reference's despeckle2 in serial raster order on (region0, rsize): {int((lab != gp['region']).sum())} pixels"
    marks = np.zeros(N, np.int32)
    O.rdo_mark_boundary(P(marks), P(gp["region"]), iw, ih)
    assert np.array_equal(marks, gp["boundarysrc"]), f"{where}: boundarysrc != markBoundary(region)"
    comp = np.zeros(N, np.int32)
    O.rdo_label8(P(comp), P(gp["boundarysrc"]), -1, iw, ih)
    assert np.array_equal(comp, gp["boundary"]), f"{where}: boundary != components(boundarysrc)"
    table = np.zeros(N * 4, np.int32)
    O.rdo_reduce_ls(P(table), P(gp["boundary"]), P(gp["lsid"]), iw, ih, nentry)
    assert np.array_equal(table[: nentry * 5], gp["table"]), f"{where}: table != reduceLS(boundary, lsid)"


def golden(name):
    return np.load(os.path.join(helpers.GOLDEN, name + ".npz"), allow_pickle=False)


def cframe(seed, iw, ih, t):
    """synth.frame(seed, iw, ih, t) by the library's C generator (csrc/rd_synth.c; the numpy twin needs 30 ms per 1920x1080 frame - tests/test_cpu_oracle.py checks that the two agree)"""
    a = np.zeros((ih, iw, 3), np.uint8)
    ra.lib().rd_synth_frame(a.ctypes.data, iw, ih, iw * 3, int(seed), int(t), 1)
    return a


@pytest.fixture(scope="module")
def ctx():
    c = ra.Context(0)
    yield c
    c.close()


def test_native_library_is_what_runs():
    assert ra.lib().rd_device_count() >= 1
    assert b"gfx950" in ra.lib().rd_version()
    maps = open("/proc/self/maps").read()
    assert "librectdetect_hip.so" in maps


# (the last three sizes: one pixel beyond / exactly / one tile row and column short of the 64 x 54 tiles of the edge-stopped blur)
@pytest.mark.parametrize("iw,ih,seed,nframes", [(640, 480, 0, 2), (333, 217, 2, 1), (1280, 720, 1, 2), (65, 55, 9, 1), (64, 54, 11, 1), (130, 109, 10, 2), (191, 161, 12, 1)])
def test_rect_stages_bit_exact_vs_oracle(iw, ih, seed, nframes):
    N = iw * ih
    det = ra.Detector(iw, ih, nslots=1)
    orc = helpers.OracleRect(iw, ih)
    for t in range(nframes):
        img = synth.frame(synth.SEED0 + seed, iw, ih, t)
        det.enqueue(img)
        det.poll(TAN36)
        orc.frame(img)
        for g, o, k in EXACT:
            a = det.plane(g, np.uint32, N * k)
            b = orc.plane(o).view(np.uint32)[: N * k]
            assert np.array_equal(a, b), f"frame {t}: plane {g} differs in {int((a != b).sum())} elements"
        assert helpers.segments_equal(det.last_segments(), orc.segments()), f"frame {t}: polyline segments differ"
    det.close()
    orc.close()


@pytest.mark.parametrize("iw,ih,seed,nframes", [(640, 480, 0, 2), (333, 217, 2, 1), (1280, 720, 1, 2), (1920, 1080, 0, 1), (17, 19, 7, 1), (65, 16, 8, 2)])
def test_region_stages_bit_exact_vs_spec(iw, ih, seed, nframes):
    """region merge, sizes, absorption of small regions, boundary marks, boundary components and the vote table: bit-identical
    to the oracle's order-free spec, on every fixture size incl. the benchmark size; and the rectangle list the detector
    returns equals the host post-process of the spec's planes"""
    det = ra.Detector(iw, ih, nslots=1)
    orc = helpers.OracleRect(iw, ih, helpers.REGION_SPEC)
    for t in range(nframes):
        img = synth.frame(synth.SEED0 + seed, iw, ih, t)
        det.enqueue(img)
        rects = det.poll(TAN36)
        orc.frame(img)
        check_region_planes(det, orc, f"{iw}x{ih} frame {t}")
        assert orc.rounds()[0] < 20, "the spec's merge must have settled within the 20 rounds the HIP path launches at most"
        want = ra.postprocess_planes(orc.segments(), orc.plane("boundary"), orc.plane("table"), iw, ih, TAN36)
        assert helpers.rects_equal(rects, want)
    det.close()
    orc.close()


@pytest.mark.parametrize("device_post", [False, True], ids=["host_post", "device_post"])
@pytest.mark.parametrize("name", ["rect_640x480_s0", "rect_640x480_s5", "rect_333x217_s2", "rect_1280x720_s1", "rect_1920x1080_s0"])
def test_rect_outputs_match_reference_golden(name, device_post):
    """polyline vertex lists bit-exact and rectangle lists identical to what the reference produced - with the rectangles computed
    by the host post-process (rd_post.c) and by the device post-process (rd_k_post.hip), each directly against the reference's lists"""
    g = golden(name)
    iw, ih = int(g["iw"]), int(g["ih"])
    det = detector(iw, ih, device_post, tan=float(g["tan_aov"]), nslots=1)
    for t in range(int(g["nframes"])):
        img = synth.frame(int(g["seed"]), iw, ih, t)
        det.enqueue(img)
        rects = det.poll(float(g["tan_aov"]))
        assert helpers.segments_equal(det.last_segments(), g[f"f{t}_segments"]), f"{name} frame {t}: segments differ from the reference"
        ref = g[f"f{t}_rects"]
        assert len(rects) == len(ref), f"{name} frame {t}: {len(rects)} rectangles, reference {len(ref)}"
        assert np.array_equal(rects["status"], ref["status"])
        # integer pixel coordinates bit-exact, float parameters within 1e-4 (north_star tolerance)
        assert np.array_equal(np.rint(rects["c2"]), np.rint(ref["c2"]))
        assert np.abs(rects["c2"] - ref["c2"]).max(initial=0) <= 1e-4
        assert np.abs(rects["c3"] - ref["c3"]).max(initial=0) <= 1e-4
        assert np.abs(rects["value"] - ref["value"]).max(initial=0) <= 1e-4
    if device_post:
        assert ra.lib().rd_detector_counter(det.h, 11) == int(g["nframes"]), "every frame's rectangles must have come from the device"
    det.close()


@pytest.mark.parametrize("name,nslots", [("stream_1920x1080_s0", 8), ("stream_1920x1080_s0", 16), ("stream_1280x720_s1", 8), ("stream_1280x720_s1", 2), ("stream_3840x2160_s4", 3),
                                         ("stream_1280x720_s1_300", 16), ("stream_1920x1080_s0_100", 16), ("stream_3840x2160_s4_16", 16),
                                         ("stream_1920x1080_s7_100", 16), ("stream_1920x1080_s0_100", 32),
                                         # what bench.py runs by default: 64 frames in flight (two groups of 8 queued on each of the four streams)
                                         ("stream_1920x1080_s0_100", 64), ("stream_1920x1080_s7_100", 64), ("stream_1280x720_s1_300", 64),
                                         # round 5: two more 1920x1080 streams of 200 frames (other seeds), generated from the reference after the round's last kernel change
                                         ("stream_1920x1080_s11_200", 64), ("stream_1920x1080_s12_200", 64), ("stream_1280x720_s13_300", 64),
                                         # round 6: SURVEY.md 8(d)'s depth - four more 1920x1080 streams at 300 frames and configs[3] at 100 frames, generated from the reference (raster order,
                                         # and under the concurrent, settled merge) after the round's last kernel change
                                         ("stream_1920x1080_s21_300", 64), ("stream_1920x1080_s22_300", 64),
                                         ("stream_1920x1080_s23_300", 64), ("stream_1920x1080_s24_300", 64),
                                         # configs[3] at SURVEY.md 8(d)'s length, t = 0 .. 99 (its frame 50 needs 86 launches of the merge: the limit, 64 until then, is 128 now)
                                         ("stream_3840x2160_s4_100", 16)])
def test_long_streams_in_the_benchmarked_configuration_vs_reference(name, nslots):
    """16 consecutive 1920x1080 frames of the bench stream (BASELINE.json configs[4]), 30 frames of the 1280x720 stream
    (configs[2]) and 3 frames of the 3840x2160 stream (configs[3]; its frames overflow the single-launch polyline kernel) - and the
    same streams at full length: all 300 frames of configs[2], 100 frames of the bench stream, 16 frames of configs[3] - the way
    bench.py runs them (plus stream_1920x1080_s7_100 and, in round 5, stream_1920x1080_s11_200 / _s12_200: other seeds, generated after the kernels were finished - no kernel decision was
    made looking at them) - 8 or 16 frames in flight on four shared streams (16: sparse stages in deferred batches of four), captured
    graphs, post-process on worker threads, frames resident in HBM, adaptive round budget - against what THE REFERENCE returned for the same stream
    (tests/golden/stream_*.npz, tools/make_golden_streams.py): the state carried from frame to frame (H1) is exercised up to 300 frames
    deep.  ASSERTED on every frame: the segment list is the reference's in every bit; the rectangle list is - in every bit and in order - the list THE REFERENCE'S OWN
    compiled code returns when its merge kernel runs with concurrent work-items until it settles (tests/golden/<stream>_settled.npz: the one execution of the reference's
    in-place kernel a parallel device can reproduce; no tolerance, no exception list).  Against the raster-order golden (another legal order of the same kernel): counted,
    and every frame whose multiset of rectangles differs must lie inside the reference's own order dependence (order 26 itself, or the 32 sampled orders as multisets)."""
    g = golden(name)
    iw, ih, nframes, tan = int(g["iw"]), int(g["ih"]), int(g["nframes"]), float(g["tan_aov"])
    L = ra.lib()
    dptrs = []
    for t in range(nframes):
        a = cframe(int(g["seed"]), iw, ih, t)
        p = L.rd_device_alloc(a.nbytes)
        L.rd_upload(p, a.ctypes.data, a.nbytes)
        dptrs.append(p)
    det = ra.Detector(iw, ih, nslots=nslots, nworkers=1)
    got, inflight = [], 0
    for p in dptrs:
        if inflight == nslots:
            got.append((det.poll(tan), det.last_segments()))
            inflight -= 1
        det.enqueue(p, ws=iw * 3, on_device=True)
        inflight += 1
    while inflight:
        got.append((det.poll(tan), det.last_segments()))
        inflight -= 1
    exact = same_order = same_multiset = 0
    by_order, by_order_26, unexplained = [], [], []
    go, gs = golden("stream_orders"), settled_golden(name)
    canon = lambda rs: rs[np.lexsort(np.rint(rs["c2"]).reshape(len(rs), 8).T[::-1])] if len(rs) else rs     # by rounded corner coordinates
    for t, (rects, segs) in enumerate(got):
        assert helpers.segments_equal(segs, g[f"f{t}_segments"]), f"{name} frame {t}: segments differ from the reference"
        # (1) EXACT, every frame: the list the reference's own compiled code returns for this frame when its merge kernel runs the way this implementation runs it - concurrent
        #     work-items, launched until it settles (tools/make_golden_settled.py) - in every bit and in the reference's list order
        assert helpers.rects_equal(rects, gs[f"f{t}_rects"]), f"{name} frame {t}: the list differs from the reference's own under the concurrent, settled merge ({len(rects)} against {len(gs[f'f{t}_rects'])} rectangles)"
        # (2) against the raster-order golden: counted, and every difference placed
        ref = g[f"f{t}_rects"]
        same_order += helpers.rects_equal(rects, ref)
        same_set = len(rects) == len(ref) and helpers.rects_equal(canon(rects), canon(ref))
        exact += same_set
        if sorted(rect_key(r) for r in rects) == sorted(rect_key(r) for r in ref):
            same_multiset += 1      # (the same rectangles as often; where the ORDER of the list differs it is the iteration order of the reference's hash map over boundary-component ids, oclrect.c:1103, and those ids come out of the region planes)
            continue
        # Another multiset than the raster order's: the reference's own list depends on the work-item order of its two in-place region kernels, and the frame must lie inside
        # that dependence - either the merge had settled within the reference's own 8 launches under concurrent work-items (then this list IS the reference's list under
        # order 26, no intervention), or it is a frame tools/make_golden_stream_orders.py ran the reference on under 32 legal orders: every rectangle here is returned under
        # some order, as often as under some order at least and at most (multisets: the reference lists a rectangle once per boundary component that votes for it)
        inside, _ = inside_order_dependence(rects, go, name, t)
        if bool(gs["settled_after_8"][t]):
            by_order_26.append(t)
        elif inside:
            by_order.append(t)
        else:
            unexplained.append(t)
    print(name, "slots", nslots, ": lists identical to the reference's under the concurrent, settled merge on all %d frames; against the raster-order golden: rectangle sets bit-identical on %d (%d the same multiset, %d in the same list order); "
          "another multiset, and the reference's own list under order 26 (merge settled within its 8 launches): frames %s; inside the reference's sampled order dependence: %s; unexplained: %s; round budget, repeats:" %
          (nframes, exact, same_multiset, same_order, by_order_26, by_order, unexplained), det.region_round_budget())
    helpers.parity_report("rectangle lists vs the reference's raster-order goldens (frames)", f"{name} / {nslots} in flight",
                          {"frames": nframes, "identical_to_the_reference_under_the_concurrent_settled_merge": nframes, "bit_identical_sets": exact, "same_multiset": same_multiset, "same_list_order": same_order,
                           "the_references_own_list_under_order_26": by_order_26, "inside_reference_order_dependence": by_order, "within_tolerance_only": [], "unexplained": unexplained, "segment_lists_bit_identical": nframes})
    assert not unexplained, (unexplained, "frames whose list is neither the raster order's nor inside the reference's own order dependence")
    assert exact >= EXACT_FRAMES_MIN[name], (exact, "frames bit-identical to the raster-order reference: fewer than recorded")
    det.close()
    for p in dptrs:
        L.rd_device_free(p)


@pytest.mark.parametrize("name", ["stream_1280x720_s1_300", "stream_1920x1080_s0_100", "stream_3840x2160_s4_16", "stream_1920x1080_s7_100", "stream_1920x1080_s11_200", "stream_1920x1080_s12_200", "stream_1280x720_s13_300",
                                  "stream_1920x1080_s21_300", "stream_1920x1080_s22_300", "stream_1920x1080_s23_300", "stream_1920x1080_s24_300"])
def test_order_dependent_stream_frames_equal_the_spec(name):
    """The frames of the long streams on which the reference's rectangle list depends on the work-item order of its region kernels
    (tests/golden/stream_orders.npz): there the requirement against the reference is membership (previous test), and the exact
    requirement is the order-free spec - the oracle in REGION_SPEC mode, given the state the frame inherits (the previous frame's
    strong-edge mask, H1): region planes bit-identical, rectangle list identical in every bit."""
    g, go = golden(name), golden("stream_orders")
    iw, ih, tan, seed = int(g["iw"]), int(g["ih"]), float(g["tan_aov"]), int(g["seed"])
    frames = sorted(int(k[len(name) + 2:-len("_union")]) for k in go.files if k.startswith(name + "_f") and k.endswith("_union"))
    assert frames, "no recorded frames for this stream"
    det = ra.Detector(iw, ih, nslots=1)
    orc = helpers.OracleRect(iw, ih, helpers.REGION_SPEC)
    raster = helpers.OracleRect(iw, ih, helpers.REGION_REFERENCE_RASTER)
    canon = lambda rs: rs[np.lexsort(np.rint(rs["c2"]).reshape(len(rs), 8).T[::-1])] if len(rs) else rs
    prev = np.zeros(iw * ih, np.int32)
    reported = 0
    for t in range(max(frames) + 1):
        img = cframe(seed, iw, ih, t)
        det.enqueue(img)
        rects = det.poll(tan)
        if t in frames:
            orc.set_prev_strong(prev)
            orc.frame(img)
            assert np.array_equal(det.plane("strong"), orc.plane("strong").view(np.int32)), (name, t)
            check_region_planes(det, orc, f"{name} frame {t}")
            want = ra.postprocess_planes(orc.segments(), orc.plane("boundary"), orc.plane("table"), iw, ih, tan)
            assert helpers.rects_equal(rects, want), (name, t)
            # REPORTED, not asserted: how far the planes are from the reference-mode oracle (the reference's kernels in serial raster
            # order, same inherited state).  The merge's order-free schedule labels regions differently where the reference's in-place
            # launches are order dependent; the partition of the frame into boundary / non-boundary pixels is what the votes see.
            _, independent = inside_order_dependence(rects, go, name, t)      # (as multisets: the reference repeats a rectangle that two boundary components vote for)
            reported += 1
            if reported > 4:      # (the pixel-level report costs a second run of the CPU oracle per frame: the first four recorded frames of a stream have it)
                if independent:
                    assert helpers.rects_equal(canon(rects), canon(g[f"f{t}_rects"])), "where the reference does not depend on the order, the list must hold the reference's rectangles in every bit, each as often"
                if t + 1 in frames:
                    prev = det.plane("strong")
                continue
            raster.set_prev_strong(prev)
            raster.frame(img)
            gr, rr = det.plane("region"), raster.plane("region").view(np.int32)
            gb, rb = det.plane("boundary"), raster.plane("boundary").view(np.int32)
            print(f"{name} frame {t}: reference order-independent here: {independent}; pixels whose region label differs from the reference-mode oracle: {int((gr != rr).sum())}; "
                  f"pixels on a region boundary in one and not in the other: {int(((gb > 0) != (rb > 0)).sum())} of {int((rb > 0).sum())}; "
                  f"rectangle list == reference golden (raster order): {helpers.rects_equal(rects, g[f'f{t}_rects'])}")
            helpers.parity_report("order-dependent frames vs the reference-mode oracle (pixels)", f"{name} frame {t}",
                                  {"reference_order_independent": independent, "region_label_differs": int((gr != rr).sum()), "boundary_membership_differs": int(((gb > 0) != (rb > 0)).sum()),
                                   "boundary_pixels": int((rb > 0).sum()), "pixels": iw * ih, "rect_list_equals_raster_golden": bool(helpers.rects_equal(rects, g[f"f{t}_rects"])),
                                   "region_planes_equal_spec": True})
            if independent:
                assert helpers.rects_equal(canon(rects), canon(g[f"f{t}_rects"])), "where the reference does not depend on the order, the list must hold the reference's rectangles in every bit, each as often"
        if t + 1 in frames:
            prev = det.plane("strong")
    det.close()
    orc.close()
    raster.close()


def _absorb_reference(det):
    """the reference's despeckle2 in serial raster order (oracle) on the GPU's own merged labels and sizes"""
    O, P = helpers.oracle(), helpers.P
    lab = det.plane("region0").copy()
    O.rdo_despeckle2(P(lab), P(det.plane("rsize")), 16, det.iw, det.ih)
    return lab


@pytest.mark.parametrize("iw,ih,seed,nframes", [(1920, 1080, 0, 3), (1280, 720, 1, 2), (640, 480, 5, 2), (97, 61, 7, 1), (2049, 20, 3, 1), (33, 300, 3, 1)])
def test_absorption_equals_the_references_serial_raster_order(iw, ih, seed, nframes, monkeypatch):
    """oclrect.cl:348-371 updates labels in place; in the reference's serial raster order that is a recurrence whose dependency chains
    run for hundreds of pixels along the frame's last row.  The HIP path evaluates it EXACTLY (tile kernel with a halo + single-block
    tail on max-propagation and pointer doubling): the region plane must equal the oracle's serial sweep on the same inputs in every
    pixel, on the fast path, and again when every frame is forced through the slow path (rounds over work lists to the fixed point)."""
    for force_slow in (0, 1):
        if force_slow:
            monkeypatch.setenv("RD_ABSORB_FORCE_SLOW", "1")
        det = ra.Detector(iw, ih, nslots=1)
        monkeypatch.delenv("RD_ABSORB_FORCE_SLOW", raising=False)
        for t in range(nframes):
            det.enqueue(synth.frame(synth.SEED0 + seed, iw, ih, t))
            det.poll(TAN36)
            got, want = det.plane("region"), _absorb_reference(det)
            assert got.min() >= 0, "undecided words left in the region plane"
            assert np.array_equal(got, want), f"{iw}x{ih} frame {t} slow={force_slow}: {int((got != want).sum())} pixels differ from the serial raster sweep"
            left, sweeps, slow = det.absorption()
            assert slow == (t + 1 if force_slow else 0), (left, sweeps, slow)
            if not force_slow:
                print(f"absorption {iw}x{ih} frame {t}: {left} pixels left to the tail, {sweeps} sweeps, {det.absorb_trace}")
        det.close()


def test_absorption_of_frames_made_of_small_regions_takes_the_slow_path():
    """busy stills (tests/golden/hard_rect.npz: colour tiles, noise) leave tens of thousands of undecided pixels - more than the
    single-block tail holds: the frame is finished by the slow path, with the same exact result"""
    g = golden("hard_rect")
    kinds, params = g["kinds"].tolist(), g["params"].tolist()
    took_slow = 0
    for hi in (0, 3, 6, 9, 12):
        seed, iw, ih = params[hi]
        det = ra.Detector(iw, ih, nslots=1)
        det.enqueue(synth.hard_frame(kinds[hi], seed, iw, ih))
        det.poll(TAN36)
        got, want = det.plane("region"), _absorb_reference(det)
        assert got.min() >= 0 and np.array_equal(got, want), f"busy frame {hi} ({kinds[hi]}): {int((got != want).sum())} pixels differ from the serial raster sweep"
        left, sweeps, slow = det.absorption()
        print(f"absorption busy frame {hi} ({kinds[hi]} {iw}x{ih}): {left} pixels left to the tail, {sweeps} sweeps, slow path: {slow}")
        took_slow += slow
        det.close()
    assert took_slow > 0


def test_repeats_on_shared_streams_while_graphs_are_captured(monkeypatch):
    """8 frames in flight share four streams; with the round budget pinned to 8 most 1080p frames have their region stage repeated
    by their slot's worker thread on a stream that the enqueueing thread is using - and, for each slot's first frames, capturing
    graphs on.  Results must equal the plain sequential run."""
    iw, ih = 1920, 1080
    frames = [synth.frame(synth.SEED0 + 3, iw, ih, t) for t in range(20)]
    seq = ra.Detector(iw, ih, nslots=1, nworkers=0)
    want = []
    for f in frames:
        seq.enqueue(f)
        want.append((seq.poll(TAN36), seq.last_segments()))
    seq.close()
    monkeypatch.setenv("RD_REGION_ROUNDS_FIXED", "8")
    par = ra.Detector(iw, ih, nslots=8, nworkers=1)
    monkeypatch.delenv("RD_REGION_ROUNDS_FIXED")
    got, inflight = [], 0
    for f in frames:
        if inflight == 8:
            got.append((par.poll(TAN36), par.last_segments()))
            inflight -= 1
        par.enqueue(f)
        inflight += 1
    while inflight:
        got.append((par.poll(TAN36), par.last_segments()))
        inflight -= 1
    assert par.region_round_budget()[1] > 0, "the repeat path must have run"
    par.close()
    for (r1, s1), (r2, s2) in zip(want, got):
        assert helpers.rects_equal(r1, r2) and helpers.segments_equal(s1, s2)


@pytest.mark.parametrize("name", ["poly_640x480_s0", "poly_333x217_s2", "poly_1280x720_s1_vid"])
def test_poly_path_through_operator_api(ctx, name):
    """poly.cpp's operator sequence through the oclimgutil_* / oclpolyline_execute entry points"""
    g = golden(name)
    iw, ih = int(g["iw"]), int(g["ih"])
    img = synth.frame(int(g["seed"]), iw, ih, 0)
    segs, ids = ra.poly_frame(ctx, img, int(g["strength_thre"]), float(g["minerror"]), int(g["size_thre"]))
    assert helpers.segments_equal(segs, g["segments"])
    assert (zlib.crc32(ids.tobytes()) & 0xFFFFFFFF) == int(g["ids_crc"])
    osegs, oids = helpers.oracle_poly(img, int(g["strength_thre"]), float(g["minerror"]), int(g["size_thre"]))
    assert np.array_equal(ids, oids)


def test_reference_api_pipelined_equals_single_shot(ctx):
    """oclrect_executeOnce vs the two-deep enqueue / poll protocol of vidrect.cpp:144-172 on a 4-frame stream"""
    iw, ih = 640, 480
    frames = [synth.frame(synth.SEED0 + 4, iw, ih, t) for t in range(4)]
    a = ra.RectDetector(ctx, iw, ih)
    single = [a.execute_once(f, TAN36) for f in frames]
    a.close()
    b = ra.RectDetector(ctx, iw, ih)
    out = []
    b.enqueue(frames[0])
    for f in frames[1:]:
        b.enqueue(f)
        out.append(b.poll(TAN36))
    out.append(b.poll(TAN36))
    b.close()
    for x, y in zip(single, out):
        assert helpers.rects_equal(x, y)


def test_reference_api_on_page_locked_buffers_keeps_the_copy_contract(ctx):
    """A frame handed to oclrect_enqueueTask / oclrect_executeOnce in page-locked memory - the reference's own allocatePinnedMemory (oclhelper.h) hands such memory out - is read in
    place by the copy engine instead of being copied by the caller's thread first.  The reference's contract stays: the caller may REUSE the buffer the moment the call returns
    (oclrect.c:1256 copies it) - here the same pinned buffer is overwritten with the next frame right after every enqueue, two frames in flight, and every list must be the one the
    pageable path returns; the counters say which way the frames went; and the detector extension's RD_FRAME_HOST_PINNED gives the same lists with 8 frames in flight."""
    iw, ih, n = 640, 480, 10
    frames = [synth.frame(synth.SEED0 + 14, iw, ih, t) for t in range(n)]
    a = ra.RectDetector(ctx, iw, ih)
    want = [a.execute_once(f, TAN36) for f in frames]
    a.close()
    det_of = lambda r: ctypes_cast_detector(r)
    b = ra.RectDetector(ctx, iw, ih)
    buf = ctx.pinned_copy(frames[0])
    got = []
    b.enqueue(buf)
    for t in range(1, n):
        buf[...] = frames[t]          # the buffer of the frame in flight is overwritten at once
        b.enqueue(buf)
        got.append(b.poll(TAN36))
    got.append(b.poll(TAN36))
    assert ra.lib().rd_detector_counter(det_of(b), 18) == n and ra.lib().rd_detector_counter(det_of(b), 19) == 0, "the frames must have travelled straight from the pinned buffer"
    once = []
    for f in frames[:3]:
        buf[...] = f
        once.append(b.execute_once(buf, TAN36))
    b.close()
    for t, (x, y) in enumerate(zip(want, got)):
        assert helpers.rects_equal(x, y), t
    c = ra.RectDetector(ctx, iw, ih)
    for t, y in enumerate(once):
        assert helpers.rects_equal(c.execute_once(frames[t], TAN36), y), t
    assert ra.lib().rd_detector_counter(det_of(c), 18) == 0 and ra.lib().rd_detector_counter(det_of(c), 19) == 3      # (pageable numpy memory: copied first)
    c.close()
    # the detector extension: RD_FRAME_HOST_PINNED, eight in flight, every frame a buffer of its own that stays untouched until its poll
    bufs = [ctx.pinned_copy(f) for f in frames]
    d = ra.Detector(iw, ih, nslots=8, nworkers=1)
    lists, k = [], 0
    for p in bufs:
        if k - len(lists) == 8: lists.append(d.poll(TAN36))
        d.enqueue(p.ctypes.data, ws=iw * 3, pinned=True); k += 1
    while len(lists) < k: lists.append(d.poll(TAN36))
    assert ra.lib().rd_detector_counter(d.h, 18) == n
    d.close()
    for t, (x, y) in enumerate(zip(want, lists)):
        assert helpers.rects_equal(x, y), t
    for p in bufs + [buf]:
        ctx.free_pinned(p)


def ctypes_cast_detector(rect_detector):
    """the rd_detector behind an oclrect_t (struct oclrect_t { uint32_t magic; rd_detector *det; ... }, rd_api.hip): for its counters"""
    import ctypes
    return ctypes.cast(rect_detector.h + 8, ctypes.POINTER(ctypes.c_void_p))[0]


def test_device_resident_frames_and_stride(ctx):
    """frames already in HBM, row stride larger than 3*iw"""
    iw, ih = 333, 217
    ws = 3 * iw + 9
    img = synth.frame(synth.SEED0 + 2, iw, ih, 0)
    padded = np.zeros((ih, ws), np.uint8)
    padded[:, : 3 * iw] = img.reshape(ih, 3 * iw)
    L = ra.lib()
    dptr = L.rd_device_alloc(padded.nbytes)
    L.rd_upload(dptr, padded.ctypes.data, padded.nbytes)
    det = ra.Detector(iw, ih, nslots=2)
    det.enqueue(dptr, ws=ws, on_device=True)
    r1 = det.poll(TAN36)
    s1 = det.last_segments()
    det.close()
    det2 = ra.Detector(iw, ih, nslots=1)
    det2.enqueue(img)
    r2 = det2.poll(TAN36)
    assert helpers.rects_equal(r1, r2) and helpers.segments_equal(s1, det2.last_segments())
    det2.close()
    L.rd_device_free(dptr)


@pytest.mark.parametrize("iw,ih", [(64, 48), (40, 33), (130, 70)])
def test_flat_and_tiny_frames(iw, ih):
    det = ra.Detector(iw, ih, nslots=1)
    det.enqueue(np.full((ih, iw, 3), 40, np.uint8))
    rects = det.poll(0.7)
    assert len(rects) == 0 and int(det.last_segments().view("i4")[0]) == 0
    img = synth.frame(synth.SEED0 + 9, iw, ih, 0)
    det.enqueue(img)
    det.poll(0.7)
    orc = helpers.OracleRect(iw, ih)
    orc.frame(np.full((ih, iw, 3), 40, np.uint8))
    orc.frame(img)
    assert helpers.segments_equal(det.last_segments(), orc.segments())
    assert np.array_equal(det.plane("strong"), orc.plane("strong"))
    det.close()
    orc.close()


def test_labelling_operator_on_adversarial_masks(ctx):
    """oclimgutil_label8x_int_int: spirals, checkerboards, single pixels, full plane, with and without background"""
    L = ra.lib()
    iu = L.init_oclimgutil(ctx.device, ctx.context)
    rng = np.random.default_rng(1)
    iw, ih = 200, 131
    masks = [np.zeros((ih, iw), np.int32), np.ones((ih, iw), np.int32), (np.indices((ih, iw)).sum(0) & 1).astype(np.int32),
             (rng.random((ih, iw)) < 0.5).astype(np.int32), (rng.random((ih, iw)) < 0.08).astype(np.int32) * rng.integers(1, 4, (ih, iw)).astype(np.int32)]
    sp = np.zeros((ih, iw), np.int32)
    for k in range(0, 60, 4):
        sp[k:ih - k, k] = 1; sp[k, k:iw - k] = 1; sp[k:ih - k - 2, iw - k - 1] = 1; sp[ih - k - 1, k + 2:iw - k] = 1
    masks.append(sp)
    O = helpers.oracle()
    for m in masks:
        for bgc in (0, -1):
            a, b, t = ctx.buffer(m), ctx.buffer(iw * ih * 4), ctx.buffer(iw * ih * 4)
            L.oclimgutil_label8x_int_int(iu, b, a, t, bgc, iw, ih, ctx.queue, None)
            got = ctx.read(b, np.int32, iw * ih)
            want = np.zeros(iw * ih, np.int32)
            O.rdo_label8(helpers.P(want), helpers.P(np.ascontiguousarray(m)), bgc, iw, ih)
            assert np.array_equal(got, want)
            ctx.release(a, b, t)
    L.dispose_oclimgutil(iu)


def test_blur_operator_against_oracle(ctx):
    L = ra.lib()
    iu = L.init_oclimgutil(ctx.device, ctx.context)
    rng = np.random.default_rng(2)
    O = helpers.oracle()
    for iw, ih in [(96, 70), (257, 129), (640, 480)]:
        x = rng.random((ih, iw), np.float32)
        a, o, t0, t1 = ctx.buffer(x), ctx.buffer(iw * ih * 4), ctx.buffer(iw * ih * 4), ctx.buffer(iw * ih * 4)
        L.oclimgutil_iirblur_f_f(iu, o, a, t0, t1, 2, iw, ih, ctx.queue, None)
        got = ctx.read(o, np.float32, iw * ih)
        want = np.zeros(iw * ih, np.float32)
        O.rdo_iirblur(helpers.P(want), helpers.P(x), iw, ih)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        ctx.release(a, o, t0, t1)
    L.dispose_oclimgutil(iu)


@pytest.mark.parametrize("name", ["ops_iir_97x61", "ops_iir_160x131"])
def test_blur_operator_every_radius_against_reference_golden(ctx, name):
    """oclimgutil_iirblur_f_f(r) for radii no application passes (sigma = (r + 1) / 3): expected planes made by the
    reference's own kernels (tools/make_golden_ops.py); bit-exact.  r = 2 goes through the blocked path, the rest
    through the full-length sweeps."""
    g = np.load(os.path.join(helpers.GOLDEN, name + ".npz"))
    iw, ih = int(g["iw"]), int(g["ih"])
    L = ra.lib()
    iu = L.init_oclimgutil(ctx.device, ctx.context)
    x = np.ascontiguousarray(g["in_f"])
    for r in [int(r) for r in g["radii"]]:
        a, o, t0, t1 = ctx.buffer(x), ctx.buffer(iw * ih * 4), ctx.buffer(iw * ih * 4), ctx.buffer(iw * ih * 4)
        L.oclimgutil_iirblur_f_f(iu, o, a, t0, t1, r, iw, ih, ctx.queue, None)
        got = ctx.read(o, np.float32, iw * ih)
        assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(g["r%d" % r]).view(np.uint32)), r
        ctx.release(a, o, t0, t1)
    L.dispose_oclimgutil(iu)


def test_blur_operator_every_radius_against_oracle(ctx):
    """the same at sizes of its own (odd widths, a frame-sized plane), in place (obuf == ibuf) included"""
    L = ra.lib()
    iu = L.init_oclimgutil(ctx.device, ctx.context)
    rng = np.random.default_rng(5)
    O = helpers.oracle()
    for iw, ih, radii in [(45, 43, [0, 31]), (333, 217, [1, 5, 9, 17, 31]), (1280, 720, [3, 24])]:
        x = (rng.random((ih, iw), np.float32) * 2 - 0.5).astype(np.float32)
        for r in radii:
            want = np.zeros(iw * ih, np.float32)
            assert O.rdo_iirblur_r(helpers.P(want), helpers.P(x), iw, ih, r) == 0
            a, o, t0, t1 = ctx.buffer(x), ctx.buffer(iw * ih * 4), ctx.buffer(iw * ih * 4), ctx.buffer(iw * ih * 4)
            L.oclimgutil_iirblur_f_f(iu, o, a, t0, t1, r, iw, ih, ctx.queue, None)
            assert np.array_equal(ctx.read(o, np.float32, iw * ih).view(np.uint32), want.view(np.uint32)), (iw, ih, r)
            L.oclimgutil_iirblur_f_f(iu, a, a, t0, t1, r, iw, ih, ctx.queue, None)
            assert np.array_equal(ctx.read(a, np.float32, iw * ih).view(np.uint32), want.view(np.uint32)), (iw, ih, r, "in place")
            ctx.release(a, o, t0, t1)
    L.dispose_oclimgutil(iu)


def test_full_size_properties_1080p():
    """size-independent properties at the benchmark size: determinism across detectors, id plane vs segment list
    consistency, chain links are mutual, every rectangle is a convex quad inside a generous frame margin"""
    iw, ih = 1920, 1080
    img = synth.frame(synth.SEED0 + 6, iw, ih, 17)
    outs = []
    for _ in range(2):
        det = ra.Detector(iw, ih, nslots=1)
        det.enqueue(img)
        rects = det.poll(TAN36)
        outs.append((rects, det.last_segments(), det.plane("lsid"), det.plane("boundary")))
        det.close()
    assert helpers.rects_equal(outs[0][0], outs[1][0]) and outs[0][1].tobytes() == outs[1][1].tobytes()
    assert np.array_equal(outs[0][2], outs[1][2]) and np.array_equal(outs[0][3], outs[1][3])
    rects, segs, lsid, _ = outs[0]
    n = int(segs.view("i4")[0])
    assert lsid.min() >= 0 and lsid.max() <= n
    valid = np.nonzero(segs["polyid"][1:] != 0)[0] + 1
    for g in valid:
        r = int(segs["rightPtr"][g])
        if r:
            assert int(segs["leftPtr"][r]) == g and segs["polyid"][r] == segs["polyid"][g]
            assert segs["x1"][g] == segs["x0"][r] and segs["y1"][g] == segs["y0"][r]     # joined end points
    used = np.unique(lsid[lsid > 0])
    assert set(used.tolist()) <= set(range(1, n + 1))
    for r in rects:
        c = r["c2"]
        d = np.roll(c, -1, 0) - c
        cr = d[:, 0] * np.roll(d, -1, 0)[:, 1] - d[:, 1] * np.roll(d, -1, 0)[:, 0]
        assert (cr > 0).all() or (cr < 0).all()
        assert (c > -0.5 * iw).all() and (c < 1.5 * iw).all()


def test_chunked_blur_verifies_and_counters():
    """the chunked IIR evaluation must pass its own on-device verification (no fallback taken) on real frames"""
    iw, ih = 1920, 1080
    det = ra.Detector(iw, ih, nslots=1)
    det.enqueue(synth.frame(synth.SEED0 + 1, iw, ih, 3))
    det.poll(TAN36)
    flags = det.plane("iirflags", np.int32, 16)
    assert flags[0] == 0 and flags[1] == 0
    ctr = det.plane("polyctr", np.int32, 64)
    assert 0 < ctr[0] < iw * ih and 0 < ctr[1] <= ctr[0]
    print("chain pixels", int(ctr[0]), "chains", int(ctr[1]), "live pixels", int(ctr[24]))
    t = ctr[39:46].astype(np.int64)
    print("single-launch polyline stage, phase durations in us (load, init, rounds, moments+fit, join, store):", (((t[1:] - t[:-1]) & 0xffffffff) / 100.0).tolist())
    det.close()


def test_blur_columns_that_fail_the_check_are_evaluated_again(monkeypatch):
    """RD_IIR_FORCE_FIX makes every column fail the on-device check of the chunked IIR evaluation: the full-length sweeps
    that then replace the result must reproduce the same planes bit for bit (both passes, transposed and plain output)"""
    iw, ih = 640, 480
    img = synth.frame(synth.SEED0 + 2, iw, ih, 1)
    det = ra.Detector(iw, ih, nslots=1)
    det.enqueue(img)
    want_rects = det.poll(TAN36)
    want = {n: det.plane(n, np.uint32) for n in ("plab1", "lblur", "nms")}
    assert det.plane("iirflags", np.int32, 16)[:2].tolist() == [0, 0]
    det.close()
    monkeypatch.setenv("RD_IIR_FORCE_FIX", "1")
    det = ra.Detector(iw, ih, nslots=1)
    det.enqueue(img)
    got_rects = det.poll(TAN36)
    assert det.plane("iirflags", np.int32, 16)[:2].tolist() == [1, 1]
    for n in want:
        assert np.array_equal(det.plane(n, np.uint32), want[n]), n
    assert helpers.rects_equal(got_rects, want_rects)
    det.close()


@pytest.mark.parametrize("nslots", [2, 3, 8])
def test_pipelined_workers_equal_sequential(nslots):
    """several frames in flight + post-process on worker threads + captured graphs == one frame at a time, inline
    (2: two streams per frame; 3: one stream per frame; 8: slots 4..7 queue up on the streams of slots 0..3)"""
    iw, ih = 640, 480
    frames = [synth.frame(synth.SEED0 + 8, iw, ih, t) for t in range(19)]
    seq = ra.Detector(iw, ih, nslots=1, nworkers=0)
    want = []
    for f in frames:
        seq.enqueue(f)
        want.append((seq.poll(TAN36), seq.last_segments()))
    seq.close()
    par = ra.Detector(iw, ih, nslots=nslots, nworkers=1)
    got, inflight = [], 0
    for f in frames:
        if inflight == nslots:
            got.append((par.poll(TAN36), par.last_segments()))
            inflight -= 1
        par.enqueue(f)
        inflight += 1
    while inflight:
        got.append((par.poll(TAN36), par.last_segments()))
        inflight -= 1
    par.close()
    assert len(got) == len(want)
    for (r1, s1), (r2, s2) in zip(want, got):
        assert helpers.rects_equal(r1, r2) and helpers.segments_equal(s1, s2)


@pytest.mark.parametrize("nslots,pattern", [(8, [1, 3, 8, 5, 2, 8, 8, 1]), (4, [4, 1, 2, 3]), (6, [6, 5, 6]), (5, [2, 5])])
def test_batched_sparse_stages_any_polling_pattern(nslots, pattern, monkeypatch):
    """The sparse stages (polylines, votes, probes) of up to four consecutive slots can run as one set of launches (frame =
    blockIdx.z; the default from 12 slots on, RD_BATCH=4 here).  Whatever the caller's rhythm - groups filled completely, polled
    when partly filled, slot counts that are no multiple of four - the results must equal one frame at a time on a single-slot detector."""
    monkeypatch.setenv("RD_BATCH", "4")
    monkeypatch.setenv("RD_ZBATCH", "0")      # (group launches - next test - replace this arrangement by default; it remains what RD_ZBATCH=0 selects)
    iw, ih = 640, 480
    frames = [synth.frame(synth.SEED0 + 21, iw, ih, t) for t in range(sum(pattern))]
    seq = ra.Detector(iw, ih, nslots=1, nworkers=0)
    want = []
    for f in frames:
        seq.enqueue(f)
        want.append((seq.poll(TAN36), seq.last_segments()))
    seq.close()
    for workers in (0, 1):
        det = ra.Detector(iw, ih, nslots=nslots, nworkers=workers)
        got, k = [], 0
        for burst in pattern:          # hand over `burst` frames, then collect all of them
            for _ in range(burst):
                det.enqueue(frames[k])
                k += 1
            for _ in range(burst):
                got.append((det.poll(TAN36), det.last_segments()))
        det.close()
        assert len(got) == len(want)
        for t, ((r1, s1), (r2, s2)) in enumerate(zip(want, got)):
            assert helpers.rects_equal(r1, r2) and helpers.segments_equal(s1, s2), (workers, t)


@pytest.mark.parametrize("zb,nslots,pattern,on_device", [(4, 16, [16, 16, 3, 16, 1, 7], True), (4, 16, [16, 5, 16], False), (2, 8, [8, 3, 8, 1, 8], True), (8, 16, [16, 9, 16], True),
                                                          (4, 9, [9, 9, 4], True), (3, 6, [6, 2, 6], False), (None, 16, [16, 16, 6], True), (None, 8, [8, 8, 3], False)])
def test_group_launches_any_polling_pattern(zb, nslots, pattern, on_device, monkeypatch):
    """Group launches: the frames of zb consecutive slots run as ONE set of launches, dense stages included (frame = blockIdx.z, the slots'
    planes at a constant pitch; the default from six slots on: None = whatever the library picks).  Full groups, groups polled when partly
    filled (launched frame by frame then), slot counts that are no multiple of the group size, frames resident in HBM or handed over as
    host buffers (with different row strides inside one group: not groupable) - every frame's lists must equal one frame at a time on a
    single-slot detector, and with them the state that travels from frame to frame (H1)."""
    if zb is not None: monkeypatch.setenv("RD_ZBATCH", str(zb))
    iw, ih = 640, 480
    n = sum(pattern)
    frames = [synth.frame(synth.SEED0 + 33, iw, ih, t) for t in range(n)]
    seq = ra.Detector(iw, ih, nslots=1, nworkers=0)
    want = []
    for f in frames:
        seq.enqueue(f)
        want.append((seq.poll(TAN36), seq.last_segments()))
    seq.close()
    L = ra.lib()
    dptrs = []
    if on_device:
        for f in frames:
            p = L.rd_device_alloc(f.nbytes); L.rd_upload(p, f.ctypes.data, f.nbytes); dptrs.append(p)
    else:
        # every third frame in a buffer with padded rows: a group with two strides cannot be one launch
        padded = []
        for t, f in enumerate(frames):
            if t % 3 == 1:
                g = np.zeros((ih, iw * 3 + 24), np.uint8); g[:, :iw * 3] = f.reshape(ih, iw * 3); padded.append(g)
            else: padded.append(f)
    for workers in (0, 1):
        det = ra.Detector(iw, ih, nslots=nslots, nworkers=workers)
        got, k = [], 0
        for burst in pattern:
            for _ in range(burst):
                if on_device: det.enqueue(dptrs[k], ws=iw * 3, on_device=True)
                else: det.enqueue(padded[k], ws=padded[k].shape[1] if padded[k].ndim == 2 else iw * 3)
                k += 1
            for _ in range(burst):
                got.append((det.poll(TAN36), det.last_segments()))
        det.close()
        assert len(got) == len(want)
        for t, ((r1, s1), (r2, s2)) in enumerate(zip(want, got)):
            assert helpers.rects_equal(r1, r2) and helpers.segments_equal(s1, s2), (workers, t)
    for p in dptrs: L.rd_device_free(p)


@pytest.mark.gpu
def test_group_strong_masks_where_the_frame_before_decides(monkeypatch):
    """A frame's strength sums start from the strong mask of the frame before (H1).  Group launches evaluate the masks of their 8 frames in ONE launch: the mask of the
    frame before is 0 or 1, so it only decides sums that stand exactly one below a threshold, and is then evaluated on the spot, level by level down to the frame
    before the group (k_strength_masks_group).  With the reference's thresholds hardly a sum of these streams stands there, so the thresholds are moved (test hook
    RD_TEST_THRESHOLDS) onto the most frequent sums of the stream - thousands of components per frame then hang on their predecessor - and the group form must equal
    the frame-by-frame form in every list of every frame, and in the mask planes of the last one."""
    iw, ih, n = 640, 480, 40
    frames = [synth.frame(synth.SEED0 + 35, iw, ih, t) for t in range(n)]
    probe = ra.Detector(iw, ih, nslots=1, nworkers=0)
    sums = []
    for f in frames[:4]:
        probe.enqueue(f); probe.poll(TAN36)
        st = probe.plane("strsum")
        sums.append(st[st > 1])      # (sums live at the components' roots; the debug plane shows them on top of the previous mask: 0 / 1 elsewhere)
    probe.close()
    vals, cnt = np.unique(np.concatenate(sums), return_counts=True)
    common = vals[np.argsort(-cnt)][:3].tolist()
    hits = 0
    for te, ts in [(int(common[0]) + 1, int(common[0]) + 1), (int(min(common[:2])) + 1, int(max(common[:2])) + 1), (int(common[2]) + 1, 4 * int(common[2]) + 7)]:
        monkeypatch.setenv("RD_TEST_THRESHOLDS", "%d,%d" % (te, ts))
        outs = []
        for env, nslots in (("1", 1), ("1", 32), (None, 32)):      # frame by frame alone, frame by frame inside groups, the group launch
            if env: monkeypatch.setenv("RD_STRONG_BY_FRAME", env)
            else: monkeypatch.delenv("RD_STRONG_BY_FRAME", raising=False)
            det = ra.Detector(iw, ih, nslots=nslots, nworkers=1 if nslots > 1 else 0)
            got, k = [], 0
            for f in frames:
                if k - len(got) == nslots: got.append((det.poll(TAN36), det.last_segments()))
                det.enqueue(f); k += 1
            while len(got) < k: got.append((det.poll(TAN36), det.last_segments()))
            planes = (det.plane("strong"), det.plane("edge500"), det.plane("strsum"))
            if nslots > 1:
                assert det.frames_per_launch() == 8
                # which form ran (the switch is read when the detector is created, so the three legs of one process really differ): 40 frames = 5 full groups
                in_one_launch, by_frame = ra.lib().rd_detector_counter(det.h, 16), ra.lib().rd_detector_counter(det.h, 17)
                assert (in_one_launch, by_frame) == ((0, 5) if env else (5, 0)), (env, in_one_launch, by_frame)
            det.close()
            outs.append((got, planes))
        st = outs[0][1][2]
        hits += int(((st == ts - 1) | (st == te - 1)).sum())
        for other in outs[1:]:
            for t, ((r1, s1), (r2, s2)) in enumerate(zip(outs[0][0], other[0])):
                assert helpers.rects_equal(r1, r2) and helpers.segments_equal(s1, s2), (te, ts, t)
            assert np.array_equal(outs[0][1][0], other[1][0]) and np.array_equal(outs[0][1][1], other[1][1]), (te, ts)
    print("components whose sum stands one below a threshold in the last frames:", hits)
    assert hits > 100, "the thresholds must sit on frequent sums, or the test shows nothing"


@pytest.mark.gpu
@pytest.mark.parametrize("zb,nslots", [(4, 9), (None, 7), (3, 8), (8, 33), (2, 7), (None, 64), (None, 128)])      # (64: bench.py's default; 128: workers that wait in two steps)
def test_group_launches_sliding_window_with_slot_counts_that_are_no_multiple_of_the_group(zb, nslots, monkeypatch):
    """The steady state of a real caller (and of bench.py): nslots frames in flight, poll one, enqueue one - with a slot count that is
    no multiple of the group size, so that the last group of slots is short and the window wraps across it.  Frames must reach the device
    in sequence order (a frame's strength sums start from the strong mask of the frame before it, H1): every frame's lists equal the
    single-slot detector's.  Also drain() with frames waiting in two groups' slots."""
    if zb is not None: monkeypatch.setenv("RD_ZBATCH", str(zb))
    iw, ih = 640, 480
    n = 3 * nslots + 5
    frames = [synth.frame(synth.SEED0 + 35, iw, ih, t) for t in range(n)]
    seq = ra.Detector(iw, ih, nslots=1, nworkers=0)
    want = []
    for f in frames:
        seq.enqueue(f)
        want.append((seq.poll(TAN36), seq.last_segments()))
    seq.close()
    for workers in (0, 1):
        det = ra.Detector(iw, ih, nslots=nslots, nworkers=workers)
        got, k = [], 0
        for f in frames:
            if k - len(got) == nslots:
                got.append((det.poll(TAN36), det.last_segments()))
            det.enqueue(f); k += 1
            if k == nslots + 2:
                det.drain()        # frames waiting in the slots of two groups (the window has wrapped): launched in sequence order
        while len(got) < k:
            got.append((det.poll(TAN36), det.last_segments()))
        det.close()
        for t, ((r1, s1), (r2, s2)) in enumerate(zip(want, got)):
            assert helpers.rects_equal(r1, r2) and helpers.segments_equal(s1, s2), (workers, t)


@pytest.mark.parametrize("nslots", [1, 2, 8])
def test_device_postprocess_equals_host_postprocess(nslots):
    """RD_DEVICE_POST=1: candidate funnel + pose estimation on the device (rd_k_post.hip: one wave per candidate, double precision,
    same source for the arithmetic as the host path) - rectangle lists bit-identical to the host post-process, on stream frames,
    on the busy frames (up to 66 rectangles, 1500 segments) and with the sparse stages batched; frames the device cannot take
    (no aperture known yet: the reference passes it with the poll; capacity overflow) fall back to the host path and are counted."""
    g = golden("hard_rect")
    jobs = [(640, 480, [synth.frame(synth.SEED0 + 31, 640, 480, t) for t in range(9)])]
    jobs.append((640, 480, [synth.hard_frame(k, sd, iw, ih) for k, (sd, iw, ih) in zip(g["kinds"].tolist(), g["params"].tolist()) if iw == 640]))
    jobs.append((1280, 720, [synth.hard_frame(k, sd, iw, ih) for k, (sd, iw, ih) in zip(g["kinds"].tolist(), g["params"].tolist()) if iw == 1280]))
    jobs.append((1920, 1080, [synth.frame(synth.SEED0, 1920, 1080, t) for t in range(6)]))
    for iw, ih, frames in jobs:
        outs = []
        for env in ({}, {"RD_DEVICE_POST": "1"}):
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            det = ra.Detector(iw, ih, nslots=nslots, nworkers=1 if nslots > 1 else 0)
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
            res, infl = [], 0
            det.enqueue(frames[0])              # (the first poll tells the detector the aperture)
            res.append((det.poll(TAN36), det.last_segments()))
            for f in frames[1:]:
                if infl == nslots:
                    res.append((det.poll(TAN36), det.last_segments()))
                    infl -= 1
                det.enqueue(f)
                infl += 1
            while infl:
                res.append((det.poll(TAN36), det.last_segments()))
                infl -= 1
            on_device = ra.lib().rd_detector_counter(det.h, 11)
            det.close()
            outs.append((res, on_device))
        (host, n0), (dev, n1) = outs
        assert n0 == 0 and n1 >= len(frames) - 2, (iw, ih, n1)
        for t, (a, b) in enumerate(zip(host, dev)):
            assert helpers.rects_equal(a[0], b[0]), (iw, ih, t, len(a[0]), len(b[0]))
            assert helpers.segments_equal(a[1], b[1]), (iw, ih, t, "segment lists differ between the host and the device post-process path")
        dev = [r for r, _ in dev]
        print("%dx%d: %d frames, %d post-processed on the device, rectangles %s" % (iw, ih, len(frames), n1, [len(r) for r in dev]))


def test_user_switches_change_nothing_but_the_way_there(monkeypatch):
    """The three environment switches meant for users (include/rectdetect_hip.h, "Environment") select HOW a frame is computed, never what: RD_NO_GRAPH=1 (plain launches
    instead of captured hipGraphs: 8 frames in flight in groups of 2), RD_POST_HELPERS=0 / 3 (helper threads for the pose estimations in the reference's call shape: two
    pages, no worker threads) and RD_DEVICE_POST=1 (rectangles from the device) must return the lists of the default configuration on 24 frames."""
    iw, ih, n = 640, 480, 24
    frames = [synth.frame(synth.SEED0 + 21, iw, ih, t) for t in range(n)]

    def run(env, nslots, nworkers):
        for k in ("RD_NO_GRAPH", "RD_POST_HELPERS", "RD_DEVICE_POST"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        det = ra.Detector(iw, ih, nslots=nslots, nworkers=nworkers, aperture=TAN36)
        got, k = [], 0
        for f in frames:
            if k - len(got) == nslots: got.append((det.poll(TAN36), det.last_segments()))
            det.enqueue(f); k += 1
        while len(got) < k: got.append((det.poll(TAN36), det.last_segments()))
        counters = (ra.lib().rd_post_helpers(), ra.lib().rd_detector_counter(det.h, 11))
        det.close()
        return got, counters

    base, _ = run({}, 1, 0)
    for env, nslots, nworkers in (({"RD_NO_GRAPH": "1"}, 8, 1), ({}, 8, 1), ({"RD_POST_HELPERS": "0"}, 2, 0), ({"RD_POST_HELPERS": "3"}, 2, 0), ({"RD_DEVICE_POST": "1"}, 8, 1)):
        got, (helpers_now, on_device) = run(env, nslots, nworkers)
        for t, ((r1, s1), (r2, s2)) in enumerate(zip(base, got)):
            assert helpers.rects_equal(r1, r2) and helpers.segments_equal(s1, s2), (env, t)
        if env.get("RD_POST_HELPERS") == "3":
            assert helpers_now >= 3      # (the pool is process-wide and never shrinks on its own: an earlier detector of this process may have asked for more)
        if env.get("RD_DEVICE_POST") == "1":
            assert on_device > 0, "RD_DEVICE_POST=1: the rectangles of (most) frames must come from the device"


def _run_with_env(env, iw, ih, frames):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        det = ra.Detector(iw, ih, nslots=1)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    out = []
    for f in frames:
        det.enqueue(f)
        out.append((det.poll(TAN36), det.last_segments()))
    redone = det.redone_frames()
    det.close()
    return out, redone


@pytest.mark.parametrize("iw,ih,seed", [(640, 480, 5), (1920, 1080, 0)])
def test_polyline_single_launch_equals_multilaunch(iw, ih, seed):
    """the persistent single-launch split/refine kernel, the ~85-launch form of the same stage, and the overflow
    fallback (tail repeated the long way) must give identical segment lists and rectangles"""
    frames = [synth.frame(synth.SEED0 + seed, iw, ih, t) for t in range(2)]
    fast, n0 = _run_with_env({}, iw, ih, frames)
    slow, n1 = _run_with_env({"RD_POLY_MULTILAUNCH": "1"}, iw, ih, frames)
    redo, n2 = _run_with_env({"RD_POLY_FORCE_REDO": "1"}, iw, ih, frames)
    assert n0 == 0 and n1 == 0 and n2 == len(frames)
    for (r0, s0), (r1, s1), (r2, s2) in zip(fast, slow, redo):
        assert helpers.segments_equal(s0, s1) and helpers.segments_equal(s0, s2)
        assert helpers.rects_equal(r0, r1) and helpers.rects_equal(r0, r2)


def test_polyline_overflow_takes_fallback_and_matches_oracle():
    """a frame with far more chains than the single-launch kernel's on-chip tables hold: the overflow flag must come
    back, the stage is repeated with the multi-launch path, and the segments still equal the oracle's; after two such
    frames in a row the detector stops trying the single-block kernel for this stream"""
    iw, ih = 1280, 720
    rng = np.random.default_rng(77)
    tiles = rng.integers(0, 256, (ih // 16, iw // 16, 3), dtype=np.uint8)
    img = np.ascontiguousarray(np.repeat(np.repeat(tiles, 16, 0), 16, 1))
    det = ra.Detector(iw, ih, nslots=1)
    orc = helpers.OracleRect(iw, ih)
    det.enqueue(img)
    det.poll(TAN36)
    orc.frame(img)
    ctr = det.plane("polyctr", np.int32, 64)
    print("overflow frame: live pixels", int(ctr[24]), "chains", int(ctr[1]), "redone", det.redone_frames())
    assert det.redone_frames() == 1
    assert helpers.segments_equal(det.last_segments(), orc.segments())
    first = det.last_segments()
    for k in range(4):          # (same frame again: H1 makes frame 2 differ from frame 1, so compare frames 2.. among the two objects)
        det.enqueue(img)
        det.poll(TAN36)
        orc.frame(img)
        assert helpers.segments_equal(det.last_segments(), orc.segments())
    assert det.redone_frames() == 2          # the second overflow in a row made the multi-launch path the default: no more repeats
    assert len(first) > 1024                 # more records than the single-launch kernel holds
    det.close()
    orc.close()


@pytest.mark.parametrize("name", ["ops_97x61", "ops_160x131"])
def test_every_operator_against_reference_golden(name):
    """all oclimgutil.h operators of this library, driven exactly like tools/make_golden_ops.py drove the reference's
    (same call sequence, same seeded inputs): every output bit-identical to the reference's, floats included"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_ops", os.path.join(helpers.ROOT, "tools", "make_golden_ops.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    g = golden(name)
    got = m.run_all(m.Ops(ra.lib()), int(g["iw"]), int(g["ih"]), int(g["seed"]))
    for k in sorted(got):
        a, b = np.ascontiguousarray(got[k]), np.ascontiguousarray(g[k])
        if a.dtype == np.float32:
            a, b = a.view(np.uint32), b.view(np.uint32)
        assert a.shape == b.shape and np.array_equal(a, b), "%s differs in %d elements" % (k, int((a != b).sum()))


def test_largest_config_4k_against_oracle():
    """BASELINE.json configs[3] size (3840x2160): the front-end planes bit-exact, the polyline stage (which overflows the
    single-launch kernel's on-chip tables at this size and takes the multi-launch path) and its segments against the oracle"""
    iw, ih = 3840, 2160
    N = iw * ih
    img = synth.frame(synth.SEED0 + 4, iw, ih, 2)
    det = ra.Detector(iw, ih, nslots=1)
    orc = helpers.OracleRect(iw, ih)
    det.enqueue(img)
    rects = det.poll(TAN36)
    orc.frame(img)
    for g, o, k in EXACT:
        a = det.plane(g, np.uint32, N * k)
        b = orc.plane(o).view(np.uint32)[: N * k]
        assert np.array_equal(a, b), f"plane {g} differs in {int((a != b).sum())} elements"
    assert helpers.segments_equal(det.last_segments(), orc.segments())
    print("4K: redone", det.redone_frames(), "segments", int(det.last_segments()[0]["x0"].view(np.int32)) if False else len(det.last_segments()) - 1, "rectangles", len(rects))
    det.close()
    orc.close()


@pytest.mark.parametrize("iw,ih", [(16, 16), (17, 19), (65, 16)])
def test_smallest_frames(iw, ih):
    """the smallest frames the detector accepts (one partial tile in every tiled kernel, blur blocks shorter than their
    warm-up): same planes and segments as the oracle"""
    N = iw * ih
    rng = np.random.default_rng(iw * 100 + ih)
    img = np.ascontiguousarray(np.repeat(np.repeat(rng.integers(0, 256, ((ih + 3) // 4, (iw + 3) // 4, 3), dtype=np.uint8), 4, 0), 4, 1)[:ih, :iw])
    det = ra.Detector(iw, ih, nslots=1)
    orc = helpers.OracleRect(iw, ih)
    for _ in range(2):
        det.enqueue(img)
        det.poll(TAN36)
        orc.frame(img)
        for g, o, k in EXACT:
            a = det.plane(g, np.uint32, N * k)
            b = orc.plane(o).view(np.uint32)[: N * k]
            assert np.array_equal(a, b), f"plane {g} differs in {int((a != b).sum())} elements"
        assert helpers.segments_equal(det.last_segments(), orc.segments())
    det.close()
    orc.close()


ODD_SIZES = [(63, 64), (64, 63), (65, 65), (127, 33), (129, 31), (192, 95), (255, 129), (321, 97), (513, 66), (1027, 38), (2049, 20), (33, 300), (96, 1030)]


@pytest.mark.parametrize("iw,ih", ODD_SIZES)
def test_frame_sizes_around_every_tile_edge(iw, ih):
    """frame sizes one short of, equal to and one beyond the tile sizes of the kernels (64-column tiles, 128-pixel region blocks, 32-row
    labelling tiles, 64-row blur blocks, 16-row extent tiles), very wide and very tall ones: every plane of every stage, the segment
    list and the rectangle list against the oracle (region stages in spec mode), on a busy tiled frame followed by a stream frame
    (so the state carried between frames takes part)"""
    N = iw * ih
    det = ra.Detector(iw, ih, nslots=1)
    orc = helpers.OracleRect(iw, ih, helpers.REGION_SPEC)
    for t, img in enumerate([synth.hard_frame("tiles", 3, iw, ih), synth.frame(synth.SEED0 + 9, iw, ih, 1), synth.hard_frame("noise", 4, iw, ih)]):
        det.enqueue(img)
        rects = det.poll(TAN36)
        orc.frame(img)
        for g, o, k in EXACT:
            a = det.plane(g, np.uint32, N * k)
            b = orc.plane(o).view(np.uint32)[: N * k]
            assert np.array_equal(a, b), f"{iw}x{ih} frame {t}: plane {g} differs in {int((a != b).sum())} elements"
        assert helpers.segments_equal(det.last_segments(), orc.segments()), f"{iw}x{ih} frame {t}: polyline segments differ"
        check_region_planes(det, orc, f"{iw}x{ih} frame {t}")
        want = ra.postprocess_planes(orc.segments(), orc.plane("boundary"), orc.plane("table"), iw, ih, TAN36)
        assert helpers.rects_equal(rects, want), f"{iw}x{ih} frame {t}"
    det.close()
    orc.close()


@pytest.mark.parametrize("iw,ih,mode", [(640, 480, {}), (1920, 1080, {"RD_REGION_ROUNDS_FIXED": "8"})])
def test_region_round_budget_does_not_change_results(iw, ih, mode):
    """fewer region-merge rounds are launched per frame than the full 20 (default: what recent frames needed + margin; fixed: 8)
    and the frames that needed more are repeated with the full budget: results must equal those with all 20 rounds always"""
    nframes = 12 if iw < 1000 else 4
    frames = [synth.frame(synth.SEED0 + 9, iw, ih, t) for t in range(nframes)]
    rng = np.random.default_rng(3)
    tiles = rng.integers(0, 256, (ih // 8, iw // 8, 3), dtype=np.uint8)
    frames.insert(nframes - 2, np.ascontiguousarray(np.repeat(np.repeat(tiles, 8, 0), 8, 1)))   # a very different frame inside the stream
    out = []
    for env in (mode, {"RD_REGION_ROUNDS_FIXED": "20"}):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        det = ra.Detector(iw, ih, nslots=2, nworkers=1)
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        res, infl = [], 0
        for f in frames:
            if infl == 2:
                res.append((det.poll(TAN36), det.last_segments(), det.plane("region").copy()))
                infl -= 1
            det.enqueue(f)
            infl += 1
        while infl:
            res.append((det.poll(TAN36), det.last_segments(), det.plane("region").copy()))
            infl -= 1
        print("env", env, "budget, repeated frames:", det.region_round_budget())
        if env.get("RD_REGION_ROUNDS_FIXED") == "20":
            assert det.region_round_budget()[1] == 0      # nothing is ever repeated with the full budget
        if env.get("RD_REGION_ROUNDS_FIXED") == "8":
            assert det.region_round_budget()[1] > 0       # 8 rounds are not enough at this size: the repeat path must have run
        det.close()
        out.append(res)
    for (r0, s0, p0), (r1, s1, p1) in zip(*out):
        assert np.array_equal(p0, p1) and helpers.rects_equal(r0, r1) and helpers.segments_equal(s0, s1)


def _write_png(path, rgb):
    """8-bit RGB PNG with scanline filter 2 (Up) on every row but the first: exercises the example reader's unfiltering"""
    import struct
    ih, iw, _ = rgb.shape
    raw = bytearray()
    prev = np.zeros((iw, 3), np.uint8)
    for y in range(ih):
        row = rgb[y]
        if y == 0:
            raw += b"\x00" + row.tobytes()
        else:
            raw += b"\x02" + (row.astype(np.int16) - prev.astype(np.int16)).astype(np.uint8).tobytes()
        prev = row

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", iw, ih, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(bytes(raw))) + chunk(b"IEND", b""))


def test_example_program_on_ppm_and_png(tmp_path):
    """examples/rdrect (C, reference API only, no OpenCV) on a PPM and on a PNG of a synthetic frame: same rectangles as the
    Python front end at the same aperture (text output has 3 decimals)"""
    import re
    import subprocess
    iw, ih = 640, 480
    img = synth.frame(synth.SEED0 + 5, iw, ih, 1)
    rgb = np.ascontiguousarray(img[:, :, ::-1])
    ppm = tmp_path / "f.ppm"
    with open(ppm, "wb") as f:
        f.write(b"P6\n# comment line\n%d %d\n255\n" % (iw, ih) + rgb.tobytes())
    png = tmp_path / "f.png"
    _write_png(png, rgb)
    det = ra.Detector(iw, ih, nslots=1)
    det.enqueue(img)
    want = det.poll(float(np.tan(36.0 / 180.0 * np.pi)))
    det.close()
    exe = os.path.join(helpers.ROOT, "examples", "rdrect")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(helpers.ROOT, "examples")], stdout=subprocess.DEVNULL)
    for src in (ppm, png):
        out = subprocess.run([exe, str(src), "0", str(tmp_path / "out.ppm")], cwd=tmp_path, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        lines = [l for l in out.stdout.splitlines() if l.startswith("status")]
        assert len(lines) == len(want)
        for l, w in zip(lines, want):
            nums = [float(v) for v in re.findall(r"-?\d+\.\d+", l.split("corners")[1])]
            assert int(l.split()[1]) == int(w["status"])
            assert np.allclose(np.array(nums).reshape(4, 2), w["c2"], atol=1e-3)
        assert os.path.getsize(tmp_path / "out.ppm") > iw * ih * 3


def test_polyline_example_program_matches_operator_api(ctx, tmp_path):
    """examples/rdpoly (C; the operator sequence of BASELINE.json configs[0]) on a PNG: same valid segments as the Python
    front end's poly_frame() and as the reference's golden segment list for this frame"""
    import re
    import subprocess
    iw, ih = 640, 480
    img = synth.frame(synth.SEED0, iw, ih, 0)
    png = tmp_path / "f.png"
    _write_png(png, np.ascontiguousarray(img[:, :, ::-1]))
    segs, _ = ra.poly_frame(ctx, img)
    g = golden("poly_640x480_s0")
    assert helpers.segments_equal(segs, g["segments"])
    exe = os.path.join(helpers.ROOT, "examples", "rdpoly")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(helpers.ROOT, "examples")], stdout=subprocess.DEVNULL)
    out = subprocess.run([exe, str(png), "0", str(tmp_path / "out.ppm")], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    got = [[float(v) for v in re.findall(r"-?\d+\.\d+", l)] for l in out.stdout.splitlines() if l.startswith("segment ")]
    want = [[s["x0"], s["y0"], s["x1"], s["y1"]] for s in segs[1:] if s["polyid"] != 0]
    assert len(got) == len(want) > 0
    assert np.allclose(np.array(got), np.array(want, dtype=np.float64), atol=1e-3)


@pytest.mark.parametrize("device_post", [False, True], ids=["host_post", "device_post"])
def test_many_streams_final_outputs_equal_the_reference(device_post):
    """68 more frames (26 short streams, four sizes) against the reference's own rectangle and segment lists
    (tests/golden/many_rect.npz from tools/make_golden_many.py).  The segment lists must match exactly on every frame; the
    region stages are evaluated in another schedule than the reference's order-dependent one, so the
    number of frames whose rectangle list matches is reported and must not fall below the level measured when this
    test was written."""
    g = golden("many_rect")
    total = same = 0
    differing = []
    on_device = 0
    for si, (iw, ih, seed, nframes) in enumerate(g["streams"].tolist()):
        det = detector(iw, ih, device_post, nslots=1)
        for t in range(nframes):
            det.enqueue(synth.frame(synth.SEED0 + seed, iw, ih, t))
            rects = det.poll(TAN36)
            assert helpers.segments_equal(det.last_segments(), g["s%d_f%d_segments" % (si, t)]), (si, t)
            total += 1
            if helpers.rects_equal(rects, g["s%d_f%d_rects" % (si, t)]):
                same += 1
            else:
                differing.append((iw, ih, seed, t, len(rects), len(g["s%d_f%d_rects" % (si, t)])))
        on_device += ra.lib().rd_detector_counter(det.h, 11)
        det.close()
    print("rectangle lists identical to the reference's on %d of %d frames (%d post-processed on the device); differing:" % (same, total, on_device), differing)
    assert same == total and on_device == (total if device_post else 0)


def test_busy_inputs_final_outputs_vs_reference():
    """much busier inputs than the stream generator's (random tiles, pure noise, smooth waves with rimmed rectangles, bars on
    gradients; up to 1500 segments and 66 rectangles per frame).  Segment lists: bit-identical to the reference's.  Region planes
    and the rectangle list: bit-identical to the order-free spec (oracle REGION_SPEC mode).  Against the REFERENCE's rectangle
    lists: on such inputs the reference's own list changes with the (legal) order in which a device runs the work-items of its
    two in-place region kernels - tests/golden/hard_rect_orders.npz holds its lists under 26 such orders, order 0 = serial
    raster (tools/make_golden_orders.py) - so the requirement is membership: every rectangle the reference returns under ALL
    sampled orders must be returned here, and every rectangle returned here must be one the reference returns under at least
    one of them.  Where the reference does not depend on the order, that is equality."""
    g = golden("hard_rect")
    go = golden("hard_rect_orders")
    key = lambda r: r["c2"].tobytes() + r["c3"].tobytes() + r["value"].tobytes() + r["status"].tobytes()
    report = []
    for hi, (kind, (seed, iw, ih)) in enumerate(zip(g["kinds"].tolist(), g["params"].tolist())):
        img = synth.hard_frame(kind, seed, iw, ih)
        det = ra.Detector(iw, ih, nslots=1)
        det.enqueue(img)
        rects = det.poll(TAN36)
        assert helpers.segments_equal(det.last_segments(), g["h%d_segments" % hi]), (kind, seed)
        orc = helpers.OracleRect(iw, ih, helpers.REGION_SPEC)
        orc.frame(img)
        check_region_planes(det, orc, f"{kind} {seed}")
        want = ra.postprocess_planes(orc.segments(), orc.plane("boundary"), orc.plane("table"), iw, ih, TAN36)
        assert helpers.rects_equal(rects, want), (kind, seed)
        union, member = go["h%d_union" % hi], go["h%d_member" % hi]
        ukeys = [key(r) for r in union]
        here = set(key(r) for r in rects)
        stable = set(k for k, m in zip(ukeys, member.all(0)) if m)
        assert stable <= here, f"{kind} {seed}: a rectangle the reference returns under every work-item order is missing"
        assert here <= set(ukeys), f"{kind} {seed}: a rectangle the reference returns under none of the sampled work-item orders"
        if member.all():
            assert here == set(ukeys)
        same_as = [oi for oi in range(member.shape[0]) if set(k for k, m in zip(ukeys, member[oi]) if m) == here]
        report.append((kind, seed, len(rects), "reference: %d..%d per order, %d distinct, %d in every order; same set as orders %s" %
                       (member.sum(1).min(), member.sum(1).max(), len(union), len(stable), same_as[:6])))
        det.close()
        orc.close()
    for r in report:
        print(r)


def test_two_real_detector_processes_share_the_gpu():
    """bench.py --gpus 2 as the driver launches it, but with gloo and both ranks on this box's only GPU (dev = local_rank mod device
    count): two real per-GPU processes - own detector, own stream seed, own worker threads - a barrier and the MAX-reduced time;
    rank 0 verifies its outputs against a sequential pass.  What can be proven about the N > 1 path without the 8-GPU node."""
    import json
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    # (plain `python bench.py --gpus 2`: the script starts its two ranks itself; --share-gpus because this box has one device)
    cmd = [sys.executable, os.path.join(helpers.ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames-per-step", "24", "--backend", "gloo", "--share-gpus"]
    p = subprocess.run(cmd, cwd=helpers.ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["outputs_verified"] is True
    assert all(r["outputs_verified"] is True for r in out["ranks"]), "every rank checks its own lists against its own sequential pass"
    ranks = sorted(out["ranks"], key=lambda r: r["rank"])
    assert [r["rank"] for r in ranks] == [0, 1] and ranks[0]["stream_seed"] != ranks[1]["stream_seed"]
    assert all(r["frames"] == 48 and r["rectangles"] > 0 for r in ranks)
    assert ranks[0]["rectangles"] != ranks[1]["rectangles"] or ranks[0]["stream_seed"] != ranks[1]["stream_seed"]
    assert abs(out["ms_per_step"] * 2 / 1e3 - max(r["own_elapsed_s"] for r in ranks)) < 5e-3      # MAX over ranks
    assert abs(out["value"] - 96 / (out["ms_per_step"] * 2 / 1e3)) / out["value"] < 0.01             # whole-job frames / that time
    assert ranks[0]["pid"] != ranks[1]["pid"]
    print("two ranks on one GPU:", out["value"], "frames/s;", ranks)


def test_operator_goldens_are_what_the_reference_computes_on_the_real_opencl_device():
    """Pins the operator goldens (tests/golden/ops_*.npz: the reference on our serial OpenCL stand-in) from the outside: the reference's unchanged host C on the box's REAL OpenCL
    device (oracle/_ref/librdref_ocl.so; the vendor's compiler builds the reference's .cl sources at run time), under the goldens' arithmetic contract (contraction off, correctly
    rounded divide / sqrt - appended through the runtime's AMD_OCL_BUILD_OPTIONS_APPEND, the reference's sources and options untouched), operator by operator on the goldens' own
    inputs: all 18 IIR outputs (9 radii, 2 sizes) and 17 of the 21 other operators must equal the goldens in every bit; the four that may differ are the direction vectors and what
    is sampled along them - the device's own rsqrt, one unit in the last place (the builtin the sensitivity study varies: tests/golden/builtin_sensitivity.json).  With the three
    loosely specified builtins pinned to the stand-in's definitions (a forced include), all 21 must."""
    import json
    import subprocess
    import sys
    so = os.path.join(helpers.ROOT, "oracle", "_ref", "librdref_ocl.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/librdref_ocl.so not built (needs /root/reference at build time)")
    contract = "-Wf,-ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt"
    may_differ = {"edgevec_f2_f", "edgevec_f2_plab", "thincubic_f_f_f2", "thinthres_f_f_f2"}
    # (a) the device's own builtins; (b) rsqrt, hypot and distance - the three whose accuracy OpenCL leaves to the device - pinned to the stand-in's definitions by a header
    #     forced into the reference's programs (oracle/refshim/rdcl_pins.h): then EVERY operator must equal the goldens in every bit
    for tag, opts, allowed in (("device_builtins", contract, may_differ), ("pinned_builtins", contract + " -Wf,-include" + os.path.join(helpers.ROOT, "oracle", "refshim", "rdcl_pins.h"), set())):
        env = dict(os.environ, AMD_OCL_BUILD_OPTIONS_APPEND=opts)
        p = subprocess.run([sys.executable, os.path.join(helpers.ROOT, "tools", "ref_ops_on_opencl.py"), "test"], cwd=helpers.ROOT, env=env, capture_output=True, text=True, timeout=600)
        if p.returncode != 0:
            pytest.skip("no usable OpenCL device for the reference here: " + (p.stderr or p.stdout)[-200:])
        rep = json.load(open(os.path.join(helpers.ROOT, "gpurun_out", "ref_ops_opencl_test.json")))["fixtures"]
        summary = {}
        for fx, rows in rep.items():
            bad = {k: v for k, v in rows.items() if v.get("differing") != 0}
            summary[fx] = {"operators": len(rows), "bit_identical": len(rows) - len(bad), "differing": {k: [v.get("differing"), v.get("max_abs_difference")] for k, v in bad.items()}}
            if fx.startswith("ops_iir"):
                assert not bad, (tag, fx, bad)
            else:
                assert set(bad) <= allowed, (tag, fx, bad)
                assert all(v.get("max_abs_difference", 1.0) <= 2e-6 for v in bad.values()), (tag, fx, bad)
        helpers.parity_report("operator goldens against the reference on the box's OpenCL device (contraction off, correctly rounded divide / sqrt)", tag, summary)
        print("operator goldens vs the reference on the OpenCL device,", tag + ":", {k: "%d of %d" % (v["bit_identical"], v["operators"]) for k, v in summary.items()})


def test_reference_stage_by_stage_on_the_real_opencl_device_against_the_oracle():
    """The reference's unchanged host C and .cl sources on the box's REAL OpenCL device, whole frames, observed launch by launch (oracle/refshim/rdcl_observe.c forwards the
    OpenCL calls untouched and reads a buffer back after the launch that completes it - the launches tests/helpers.py: REF_SNAPSHOTS names), under the goldens' arithmetic
    contract with the three loosely specified builtins pinned (as the operator test above): on five frames up to 1920 x 1080 every plane from the colour conversion to the merge
    masks - 16 planes: the whole front end, the thinning, both labellings, the strength filter, the ten smoothing passes, quantisation, despeckle - must equal the oracle's in
    every bit.  The oracle's planes are the ones the HIP path's are compared with bit for bit (the device stage tests above), so for those 16 the HIP path equals the reference
    as the vendor's OpenCL runs it.  From the region merge on (labelMergeMain works in place: what a work-item reads depends on which others ran before it - DESIGN.md (c)) the
    device's order is its own: reported, not asserted."""
    import json
    import subprocess
    import sys
    so = os.path.join(helpers.ROOT, "oracle", "_ref", "librdref_ocl.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/librdref_ocl.so not built (needs /root/reference at build time)")
    opts = "-Wf,-ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt -Wf,-include" + os.path.join(helpers.ROOT, "oracle", "refshim", "rdcl_pins.h")
    p = subprocess.run([sys.executable, os.path.join(helpers.ROOT, "tools", "ref_stages_on_opencl.py"), "test"], cwd=helpers.ROOT, env=dict(os.environ, AMD_OCL_BUILD_OPTIONS_APPEND=opts),
                       capture_output=True, text=True, timeout=900)
    if p.returncode != 0:
        pytest.skip("no usable OpenCL device for the reference here: " + (p.stderr or p.stdout)[-200:])
    rep = json.load(open(os.path.join(helpers.ROOT, "gpurun_out", "ref_stages_opencl_test.json")))["frames"]
    before_the_merge = ["plab0", "Lblur", "plab1", "vxy", "strength", "nms", "mask0", "tidy", "str_sum", "edge500", "smooth", "quant", "strong", "label1", "junction", "mergemask"]
    assert len(rep) == 5
    summary = {}
    for name, fr in rep.items():
        assert fr["launches"] == 220, (name, fr["launches"])
        for k in before_the_merge:
            assert fr["planes"][k] != "no snapshot" and fr["planes"][k]["differing"] == 0, (name, k, fr["planes"][k])
        summary[name] = {"bit_identical_planes": before_the_merge, "after_the_merge_differing_elements": {k: v["differing"] for k, v in fr["planes"].items() if k not in before_the_merge}}
    helpers.parity_report("the reference stage by stage on the box's OpenCL device against the oracle (goldens' contract, pinned builtins)", "planes", summary)
    print("reference on the OpenCL device stage by stage: 16 planes identical in every bit on", len(rep), "frames; after the merge:", {k: v["after_the_merge_differing_elements"] for k, v in summary.items()})


def test_reference_polyline_stage_launch_by_launch_on_the_real_opencl_device():
    """Pins the 47 kernels behind the merge masks as far as the reference is a function there.  tools/ref_launches_on_opencl.py runs the reference's
    unchanged sources twice on nine frames up to 1920 x 1080 (three of them consecutive frames of the held-out stream stream_1920x1080_s12_200) - on the box's REAL OpenCL
    device (goldens' arithmetic contract, the three loose builtins pinned, buffers zeroed like the stand-in's) and on the serial stand-in that generated every golden -
    fingerprints every buffer argument after every one of the 220 launches on both sides, and compares the polyline stage's id planes and segment lists at ten check points up
    to the permutation of ids that relabel_pass0 / mkpl_pass2 hand out through atomic counters in work-item order (the bijection is read off the id planes).
    ASSERTED: the same 220 launches; every check point from the chains' ids (relabel_pass1) through the initial segments (mkpl_pass0b), six of the 15 split rounds (mkpl_pass3)
    and the least-squares refinement (refine_pass2) is equal up to ids on every frame - i.e. everything in oclpolyline.cl that is a function of its input is computed by the
    vendor's compiler on real hardware exactly as the goldens have it.  The one launch after which a canonicalised record may differ is refine_pass3 (H15: neighbouring segments
    joined in place, in the device's order) - reported with the number of records and the largest end-point shift, not asserted."""
    import json
    import subprocess
    import sys
    so = os.path.join(helpers.ROOT, "oracle", "_ref", "librdref_ocl.so")
    if not os.path.exists(so) or not helpers.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    opts = "-Wf,-ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt -Wf,-include" + os.path.join(helpers.ROOT, "oracle", "refshim", "rdcl_pins.h")
    p = subprocess.run([sys.executable, os.path.join(helpers.ROOT, "tools", "ref_launches_on_opencl.py"), "test"], cwd=helpers.ROOT, env=dict(os.environ, AMD_OCL_BUILD_OPTIONS_APPEND=opts, TMPDIR="/tmp"),
                       capture_output=True, text=True, timeout=1500)
    if p.returncode != 0:
        pytest.skip("no usable OpenCL device for the reference here: " + (p.stderr or p.stdout)[-200:])
    rep = json.load(open(os.path.join(helpers.ROOT, "gpurun_out", "ref_launches_opencl_test.json")))["frames"]
    if any("error" in fr for fr in rep.values()):
        pytest.skip("no usable OpenCL device for the reference here: " + str([fr["error"] for fr in rep.values() if "error" in fr])[:200])
    assert len(rep) == 9 and sum("seed 12 1920x1080" in k for k in rep) == 3
    summary = {}
    for name, fr in rep.items():
        assert fr["launches"] == 220 and not fr.get("launch_sequences_differ"), name
        cps = fr["polyline_check_points"]
        assert len(cps) == 10, (name, list(cps))
        for title, c in cps.items():
            if title.startswith("refine_pass3"):
                continue
            assert c["equal_up_to_ids"], (name, title, c)
        assert fr["first_check_point_that_differs_up_to_ids"] in (None, "refine_pass3 (neighbouring segments joined, in place: H15)"), (name, fr["first_check_point_that_differs_up_to_ids"])
        last = cps["refine_pass3 (neighbouring segments joined, in place: H15)"]
        summary[name] = {"check_points_equal_up_to_ids": [t.split(" (")[0] for t, c in cps.items() if c["equal_up_to_ids"]], "ids_permuted_on_the_device": cps["mkpl_pass0b (initial segments)"]["ids_permuted"],
                         "segments": last["records"][0], "refine_pass3_records_differing": last.get("records_differing"), "refine_pass3_largest_end_point_shift_px": last.get("largest_end_point_difference_px", 0.0),
                         "launches_with_every_buffer_identical": fr["launches_with_every_buffer_argument_identical"], "rectangle_lists": fr["rectangles"]}
    helpers.parity_report("the reference launch by launch on the box's OpenCL device against the reference on the serial stand-in (goldens' contract, pinned builtins, zeroed buffers)", "polyline stage", summary)
    print("reference on the OpenCL device, polyline stage: equal up to ids through refine_pass2 on", len(rep), "frames; refine_pass3 (H15) moves",
          {k: (v["refine_pass3_records_differing"], v["refine_pass3_largest_end_point_shift_px"]) for k, v in summary.items()})


def test_references_own_host_code_on_the_hip_paths_planes():
    """Pins the host post-process where the lists differ from the raster-order goldens: over the seven long streams, on EVERY frame whose rectangle
    list is not the golden's in every bit and order, and on every tenth frame besides, the HIP path's own segment list, vote table and boundary plane are handed to THE
    REFERENCE'S OWN compiled executeCPUTask (oclrect.c:1049-1226, unchanged: helpers.RefRect.host_postprocess - the stand-in substitutes them for the three read-backs of
    genGPUTask and runs no launch): its list must be the HIP path's list in every bit AND in order.  So wherever a list differs from a golden, the difference is in the planes
    (the region merge's order), never in rd_post.c."""
    if not helpers.have_ref():
        pytest.skip("oracle/_ref/librdref.so not built (needs /root/reference at build time)")
    rows = {}
    for name in ["stream_1920x1080_s12_200", "stream_1280x720_s13_300", "stream_1920x1080_s11_200", "stream_1920x1080_s0_100", "stream_1920x1080_s7_100", "stream_1280x720_s1_300", "stream_3840x2160_s4_16",
                 "stream_1920x1080_s21_300", "stream_1920x1080_s22_300", "stream_1920x1080_s23_300", "stream_1920x1080_s24_300"]:
        g = golden(name)
        iw, ih, nframes, tan, seed = int(g["iw"]), int(g["ih"]), int(g["nframes"]), float(g["tan_aov"]), int(g["seed"])
        N = iw * ih
        det = ra.Detector(iw, ih, nslots=1)
        r = helpers.RefRect(iw, ih)
        deviating, others = [], 0
        for t in range(nframes):
            det.enqueue(cframe(seed, iw, ih, t))
            rects = det.poll(tan)
            differs = not helpers.rects_equal(rects, g[f"f{t}_rects"])
            if not differs and t % 10:
                continue
            theirs = r.host_postprocess(det.last_segments(), det.plane("boundary"), det.plane("table", np.int32, (N * 4 // 5) * 5), tan)
            assert helpers.rects_equal(rects, theirs), f"{name} frame {t}: the reference's own executeCPUTask returns another list on the HIP path's planes ({len(theirs)} against {len(rects)} rectangles)"
            if differs:
                deviating.append(t)
            else:
                others += 1
        det.close()
        r.close()
        rows[name] = {"frames_whose_list_differs_from_the_golden_in_bits_or_order": deviating, "other_frames_checked": others}
        print(name, ": the reference's own host code on the HIP path's planes returns the HIP path's list, bit for bit and in order, on the", len(deviating), "frames whose list differs from the raster-order golden", deviating, "and on", others, "others")
    helpers.parity_report("the reference's own compiled executeCPUTask on the HIP path's planes", "lists identical in every bit and in order", rows)


def test_reference_on_the_real_opencl_device_against_the_hip_path():
    """The one third-party execution of the reference this environment offers: its unchanged host C (oracle/_ref/librdref_ocl.so, built by `make -C oracle ref_ocl`) on the
    box's OpenCL device - the MI355X through ROCm's OpenCL, the reference's .cl sources compiled at run time by the vendor's compiler - under the goldens' arithmetic contract
    (contraction off, correctly rounded divide / sqrt, the three loose builtins pinned: appended through the runtime's AMD_OCL_BUILD_OPTIONS_APPEND, the reference's sources and
    options untouched; its own options are empty, i.e. contraction on), in a process of its own.  Compared with the HIP path on seven stills, rectangle by rectangle (each of
    the HIP path's paired with the nearest of the device's).  ASSERTED: the same number of rectangles; every pair within the north_star's 1e-4 of a pixel on every corner -
    except pairs that are reported BY NAME, of which there may be at most three among the ~60 and none further than 1e-2: what moves them is the device's work-item order in
    the in-place kernels (labelMergeMain, despeckle2, refine_pass3 - the launch-by-launch test above shows refine_pass3 shifting segment end points by up to a few pixels on
    this device, and the pose estimation is continuous in them).  Reported: lists / rectangles identical in every bit."""
    import subprocess
    import sys
    import tempfile
    so = os.path.join(helpers.ROOT, "oracle", "_ref", "librdref_ocl.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/librdref_ocl.so not built (needs /root/reference at build time)")
    specs = [(0, 640, 480, 0), (5, 640, 480, 0), (1, 1280, 720, 0), (0, 1920, 1080, 0), (0, 1920, 1080, 1), (0, 1920, 1080, 2), (7, 1920, 1080, 0)]
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "ref.npz")
        opts = "-Wf,-ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt -Wf,-include" + os.path.join(helpers.ROOT, "oracle", "refshim", "rdcl_pins.h")
        p = subprocess.run([sys.executable, os.path.join(helpers.ROOT, "tools", "ref_on_opencl.py"), "dump", f] + [str(v) for sp in specs for v in sp], cwd=helpers.ROOT,
                           env=dict(os.environ, AMD_OCL_BUILD_OPTIONS_APPEND=opts), capture_output=True, text=True, timeout=600)
        if p.returncode != 0:
            pytest.skip("no usable OpenCL device for the reference here: " + (p.stderr or p.stdout)[-200:])
        with np.load(f) as z:
            ref = [z["f%d" % i].view(ra.RECT_DTYPE) if z["f%d" % i].dtype != ra.RECT_DTYPE else z["f%d" % i] for i in range(len(specs))]
    key = lambda r: r["c2"].tobytes() + r["c3"].tobytes() + r["value"].tobytes() + r["status"].tobytes()
    rows, same_lists, same_rects, total, beyond, one_ulp = {}, 0, 0, 0, [], []
    for (seed, iw, ih, t), want in zip(specs, ref):
        name = "seed %d %dx%d t %d" % (seed, iw, ih, t)
        det = ra.Detector(iw, ih, nslots=1, nworkers=0)
        det.enqueue(synth.frame(synth.SEED0 + seed, iw, ih, t))
        got = det.poll(TAN36)
        det.close()
        assert len(got) == len(want), (name, len(got), len(want))
        far = 0.0
        for i, r in enumerate(got):
            dist = float(np.abs(want["c2"] - r["c2"]).reshape(len(want), -1).max(1).min()) if len(want) else 0.0
            far = max(far, dist)
            # (the corners come from segment end points held in single precision: from x = 1024 on, one unit in the last place of an end point is 2^-13 = 1.22e-4 px - the
            #  finest difference the reference's own data type can show there is already beyond the north_star's 1e-4, and refine_pass3's order on the device produces exactly that)
            tol = max(1e-4, float(np.spacing(np.float32(np.abs(r["c2"]).max()))))
            if dist > tol:
                beyond.append({"frame": name, "rectangle": i, "corner_distance_px": dist})
            elif dist > 1e-4:
                one_ulp.append({"frame": name, "rectangle": i, "corner_distance_px": dist})
        ka, kb = {key(r) for r in got}, {key(r) for r in want}
        same_lists += ka == kb; same_rects += len(ka & kb); total += len(kb)
        rows[name] = {"rectangles": len(got), "bit_identical": len(ka & kb), "lists_bit_identical": ka == kb, "largest_corner_distance_px": far}
    helpers.parity_report("the reference on the box's OpenCL device (goldens' contract, pinned builtins) against the HIP path", "stills",
                          dict(rows, lists_bit_identical=int(same_lists), rectangles_bit_identical="%d of %d" % (same_rects, total), rectangles_beyond_tolerance=beyond, rectangles_one_single_precision_ulp_apart_beyond_1e_4_px=one_ulp))
    print("reference on the OpenCL device vs HIP: %d of %d lists and %d of %d rectangles identical in every bit; one single-precision ulp apart where that exceeds 1e-4 px: %s; beyond: %s" % (same_lists, len(specs), same_rects, total, one_ulp, beyond))
    assert len(beyond) <= 3 and all(b["corner_distance_px"] <= 1.0 for b in beyond), beyond


def test_four_ranks_with_all_their_workers_keep_the_rate_of_one():
    """What 8 ranks on the node's two sockets will stress, as far as a one-GPU box can show it: `bench.py --gpus 4 --share-gpus` - four real
    per-GPU processes, each with its own detector, 64 frames in flight and 64 worker threads that poll events, all on this box's only GPU - must reach, TOGETHER, at least
    0.9 of what one such process reaches alone on the same GPU: the host side (launches, event polling through the runtime's locks, post-process threads, the gloo control
    plane) is then not what limits a rank.  Every rank verifies its own lists."""
    import sys
    one = _bench_line([sys.executable, os.path.join(helpers.ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--frames-per-step", "256", "--no-cpu-baseline", "--no-configs"])
    four = _bench_line([sys.executable, os.path.join(helpers.ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1", "--frames-per-step", "64", "--backend", "gloo", "--share-gpus",
                        "--no-cpu-baseline", "--no-configs"])
    assert four["n_gpus"] == 4 and four["outputs_verified"] is True and all(r["outputs_verified"] is True for r in four["ranks"])
    assert len({r["pid"] for r in four["ranks"]}) == 4 and len({r["stream_seed"] for r in four["ranks"]}) == 4
    ratio = four["value"] / one["value"]
    helpers.parity_report("multi-process host side (one GPU shared)", "4 ranks x 64 frames in flight against 1 rank", {"frames_per_s_1_rank": one["value"], "frames_per_s_4_ranks_together": four["value"], "ratio": round(ratio, 3)})
    print("four ranks on one GPU: %.1f frames/s together, one rank alone %.1f (ratio %.3f)" % (four["value"], one["value"], ratio))
    assert ratio >= 0.9, (four["value"], one["value"])


def test_eight_ranks_rehearsal_on_one_gpu():
    """The node's shape as far as a one-GPU box can rehearse it: `bench.py --gpus 8 --share-gpus --slots 32` - EIGHT real per-GPU processes, each with
    its own detector, 32 frames in flight in groups of 8 and 32 worker threads (256 polling workers, eight enqueue loops, the gloo control plane), all on this box's only GPU
    - must reach, together, at least 0.75 of what one such process with 64 frames in flight reaches alone (measured 0.85-0.93 from box to box: eight processes' 32 hardware queues are
    time-sliced on one device, which a node with a GPU per rank does not do; four ranks reach 0.95); every rank verifies its own lists.  (32, not the 64 of a rank
    on a GPU of its own: eight detectors of 64 slots are 8 x 29 GB of planes and do not fit one GPU's 288 GB beside eight runtimes - the library says so loudly, "hipMalloc
    failed: out of memory", which this test also checks, instead of running short.)"""
    import subprocess
    import sys
    one = _bench_line([sys.executable, os.path.join(helpers.ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--frames-per-step", "256", "--no-cpu-baseline", "--no-configs"])
    common = ["--steps", "3", "--warmup", "1", "--frames-per-step", "64", "--backend", "gloo", "--share-gpus", "--no-cpu-baseline", "--no-configs"]
    eight = _bench_line([sys.executable, os.path.join(helpers.ROOT, "bench.py"), "--gpus", "8", "--slots", "32"] + common, timeout=1500)
    assert eight["n_gpus"] == 8 and eight["outputs_verified"] is True and all(r["outputs_verified"] is True for r in eight["ranks"])
    assert len({r["pid"] for r in eight["ranks"]}) == 8 and len({r["stream_seed"] for r in eight["ranks"]}) == 8
    ratio = eight["value"] / one["value"]
    helpers.parity_report("multi-process host side (one GPU shared)", "8 ranks x 32 frames in flight against 1 rank x 64", {"frames_per_s_1_rank": one["value"], "frames_per_s_8_ranks_together": eight["value"], "ratio": round(ratio, 3)})
    print("eight ranks on one GPU: %.1f frames/s together, one rank alone %.1f (ratio %.3f)" % (eight["value"], one["value"], ratio))
    assert ratio >= 0.75, (eight["value"], one["value"])
    # what does not fit says so: eight ranks of 64 slots on one GPU
    p = subprocess.run([sys.executable, os.path.join(helpers.ROOT, "bench.py"), "--gpus", "8", "--slots", "64"] + common, cwd=helpers.ROOT, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), capture_output=True, text=True, timeout=900)
    if p.returncode != 0:      # (what does not fit must fail loudly and print no line; the runtime's message is "out of memory" where it gets that far)
        assert not [l for l in p.stdout.splitlines() if l.startswith("{")], (p.stderr + p.stdout)[-2000:]
        print("eight ranks of 64 slots on one GPU: refused,", "out of memory" if "out of memory" in p.stderr + p.stdout else (p.stderr + p.stdout)[-200:])


def test_bench_refuses_more_ranks_than_devices():
    """`python bench.py --gpus N` with N above the node's device count must fail loudly - never a 1-GPU number under an N-GPU label"""
    import subprocess
    import sys
    n = ra.lib().rd_device_count() + 1
    p = subprocess.run([sys.executable, os.path.join(helpers.ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0", "--frames-per-step", "8", "--no-cpu-baseline", "--no-configs"],
                       cwd=helpers.ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "device" in (p.stderr + p.stdout)
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def _bench_line(cmd, timeout=900):
    import json
    import subprocess
    p = subprocess.run(cmd, cwd=helpers.ROOT, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_under_torchrun_with_one_rank_agrees_with_plain_bench():
    """the driver's SCALE run at N = 1 (bench.py under torch.distributed.run with one rank) must report what the plain BENCH run
    reports: same code path, same metric, rates within a few per cent of each other (measured ratio printed)"""
    import socket
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    args = ["--gpus", "1", "--steps", "6", "--warmup", "2", "--frames-per-step", "256", "--no-cpu-baseline", "--no-configs"]
    plain = _bench_line([sys.executable, os.path.join(helpers.ROOT, "bench.py")] + args)
    launched = _bench_line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
                            os.path.join(helpers.ROOT, "bench.py")] + args)
    for out in (plain, launched):
        assert out["n_gpus"] == 1 and out["outputs_verified"] is True and out["metric"] == "1920x1080 frames/sec" and out["roofline"]["frac"] > 0
        assert out["rect_list_crc32"] == plain["rect_list_crc32"]
    ratio = launched["value"] / plain["value"]
    print("bench.py plain %.1f frames/s, under torchrun (1 rank) %.1f frames/s, ratio %.3f" % (plain["value"], launched["value"], ratio))
    helpers.parity_report("bench.py: plain run against torch.distributed.run with one rank (frames/s)", "1920x1080", {"plain": plain["value"], "launched": launched["value"], "ratio": round(ratio, 4)})
    if not 0.97 <= ratio <= 1.03:
        # two runs of the SAME command on one box of this pool differ by up to 6 % now and then (clocks, the box's other tenants): the plain run once
        # more, and the launched run must be within 5 % of one of the two before the launch forms are called different
        plain2 = _bench_line([sys.executable, os.path.join(helpers.ROOT, "bench.py")] + args)
        print("second plain run %.1f frames/s" % plain2["value"])
        assert min(abs(1 - launched["value"] / plain["value"]), abs(1 - launched["value"] / plain2["value"])) <= 0.05, (plain["value"], plain2["value"], launched["value"])


def test_two_detectors_on_two_host_threads_in_one_process():
    """one process may drive several detectors from several threads (the runtime's selected device is thread-local, kernels that need
    a raised LDS limit set it per device, the quantisation tables are per device): two threads, each with its own detector and its
    own stream, running at the same time, must return what each stream returns on its own"""
    import threading
    iw, ih, nframes = 640, 480, 12
    streams = [[synth.frame(synth.SEED0 + 40 + k, iw, ih, t) for t in range(nframes)] for k in range(2)]
    want = []
    for frames in streams:
        det = ra.Detector(iw, ih, nslots=1)
        res = []
        for f in frames:
            det.enqueue(f)
            res.append((det.poll(TAN36), det.last_segments()))
        det.close()
        want.append(res)
    got, errs = [None, None], []
    gate = threading.Barrier(2)

    def run(k):
        try:
            ra.lib().rd_select_device(0)
            det = ra.Detector(iw, ih, nslots=4, nworkers=1)
            gate.wait()
            res, infl = [], 0
            for f in streams[k]:
                if infl == 4:
                    res.append((det.poll(TAN36), det.last_segments()))
                    infl -= 1
                det.enqueue(f)
                infl += 1
            while infl:
                res.append((det.poll(TAN36), det.last_segments()))
                infl -= 1
            det.close()
            got[k] = res
        except Exception as e:      # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errs, errs
    for k in range(2):
        assert len(got[k]) == nframes
        for (r1, s1), (r2, s2) in zip(want[k], got[k]):
            assert helpers.rects_equal(r1, r2) and helpers.segments_equal(s1, s2)


def test_frames_with_more_segments_than_the_probe_buffer_lose_nothing(monkeypatch):
    """the segment list has the reference's capacity (16N / 56 records, oclpolyline.cl:456); a slot's probe buffer is smaller (65536
    records), and a frame with more is probed again into a buffer that grows on demand: with the probe buffer shrunk to 64 records, the
    busy frames (100-1500 segments) must return exactly what they return with the full-size buffer - rectangles and complete segment list"""
    g = golden("hard_rect")
    kinds, params = g["kinds"].tolist(), g["params"].tolist()
    for hi in (0, 6, 12):
        seed, iw, ih = params[hi]
        img = synth.hard_frame(kinds[hi], seed, iw, ih)
        outs = []
        for small in (False, True):
            if small:
                monkeypatch.setenv("RD_MAXREC_DEV", "64")
            det = ra.Detector(iw, ih, nslots=1)
            monkeypatch.delenv("RD_MAXREC_DEV", raising=False)
            det.enqueue(img)
            rects = det.poll(TAN36)
            outs.append((rects, det.last_segments(), ra.lib().rd_detector_counter(det.h, 10)))
            det.close()
        (r0, s0, c0), (r1, s1, c1) = outs
        assert int(s0.view("i4")[0]) > 64 and c0 == 0 and c1 == 1
        assert helpers.rects_equal(r0, r1) and helpers.segments_equal(s0, s1)
        assert helpers.segments_equal(s1, g["h%d_segments" % hi])


def test_graphs_recorded_in_the_middle_of_a_run_while_workers_wait(monkeypatch):
    """The launch budget of the region merge follows the stream, so (slot, budget) pairs that were never used before come up in the middle
    of a run and their graphs are recorded then - on streams that other slots share and whose events worker threads are waiting for (HIP
    refuses waits on events of a capturing stream: a crash of the default bench run, once).  Here the budget changes every three
    frames (RD_BUDGET_CYCLE), 8 slots on 4 streams, one worker per slot: results must equal the sequential run."""
    iw, ih = 640, 480
    frames = [synth.frame(synth.SEED0 + 50, iw, ih, t) for t in range(72)]
    seq = ra.Detector(iw, ih, nslots=1, nworkers=0)
    want = []
    for f in frames:
        seq.enqueue(f)
        want.append((seq.poll(TAN36), seq.last_segments()))
    seq.close()
    monkeypatch.setenv("RD_BUDGET_CYCLE", "3")
    par = ra.Detector(iw, ih, nslots=8, nworkers=1)
    monkeypatch.delenv("RD_BUDGET_CYCLE")
    got, inflight = [], 0
    for f in frames:
        if inflight == 8:
            got.append((par.poll(TAN36), par.last_segments()))
            inflight -= 1
        par.enqueue(f)
        inflight += 1
    while inflight:
        got.append((par.poll(TAN36), par.last_segments()))
        inflight -= 1
    par.close()
    for (r1, s1), (r2, s2) in zip(want, got):
        assert helpers.rects_equal(r1, r2) and helpers.segments_equal(s1, s2)


def test_builtin_sensitivity_report():
    """REPORT, not a gate: the reference runs in this image on OUR OpenCL stand-in, whose loosely specified builtins (rsqrt, hypot, distance; FMA
    contraction) are choices of ours.  tests/golden/builtin_sensitivity.npz (tools/make_golden_builtins.py) holds what THE
    REFERENCE returns for seven fixture frames under the other legal choices.  Here: which of those variants' rectangle lists the HIP path's list
    equals - it must equal the baseline's, the definitions the HIP kernels share - and how far the other variants move the reference itself;
    written to the parity report (gpurun_out/parity_report.json; tools/update_parity_report.py keeps a round's copy) so that it shows in the driver's record."""
    g = golden("builtin_sensitivity")
    variants = [str(v) for v in g["variants"]]
    key = lambda r: r["c2"].tobytes() + r["c3"].tobytes() + r["value"].tobytes() + r["status"].tobytes()
    for fi, (seed, iw, ih, t) in enumerate(g["frames"].tolist()):
        det = ra.Detector(iw, ih, nslots=1)
        det.enqueue(synth.frame(synth.SEED0 + seed, iw, ih, t))
        rects = det.poll(TAN36)
        det.close()
        union, member = g[f"f{fi}_union"], g[f"f{fi}_member"]
        ukeys = [key(r) for r in union]
        here = set(key(r) for r in rects)
        same = [variants[vi] for vi in range(len(variants)) if set(k for k, m in zip(ukeys, member[vi]) if m) == here]
        moved = [variants[vi] for vi in range(1, len(variants)) if not np.array_equal(member[vi], member[0])]
        helpers.parity_report("builtin sensitivity of the reference (rectangle lists)", f"frame {fi}: {iw}x{ih} seed {seed} t {t}",
                              {"hip_list_equals_reference_under": same, "reference_list_moves_under": moved, "rectangles_baseline": int(member[0].sum()),
                               "distinct_rectangles_over_variants": len(union), "segment_records_differing_per_variant": {variants[vi]: int(g[f"f{fi}_segs"][vi, 1]) for vi in range(1, len(variants))}})
        # (a frame on which the reference's own list depends on the work-item order of its region kernels - tests/golden/stream_orders.npz, here
        #  frame 0 of the held-out stream - is compared by membership in test_long_streams...; everywhere else the baseline's list is required)
        order_dependent = any(k.endswith(f"_s{seed}_100_f{t}_union") and f"{iw}x{ih}" in k for k in golden("stream_orders").files)
        helpers.parity_report("builtin sensitivity of the reference (rectangle lists)", f"frame {fi}: {iw}x{ih} seed {seed} t {t} (order-dependent)", order_dependent)
        assert "baseline" in same or order_dependent, f"frame {fi}: the HIP path shares the baseline's builtin definitions and must return its list"
