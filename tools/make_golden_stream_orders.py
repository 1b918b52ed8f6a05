#!/usr/bin/env python3
"""The long golden streams (tests/golden/stream_*_<n>.npz, serial raster order) hold a few frames on which the REFERENCE's
rectangle list depends on the (legal) order in which a device runs the work-items of its two in-place region kernels
(rect:labelMergeMain, rect:despeckle2 - SURVEY.md H5/H6), as the busy stills of hard_rect.npz do.  For those frames this
script records what the reference itself returns under the 26 work-item orders of tools/make_golden_orders.py and six more (ORDERS below):
one detector instance runs the stream in raster order up to the frame before (so the state the reference carries from frame
to frame is the real one), then a forked copy of the process runs the frame under each order (order 0 = raster: must
reproduce the golden list).

-> tests/golden/stream_orders.npz: `<stream>_f<t>_union` = the distinct rectangles over all orders, `<stream>_f<t>_count[order, i]` = how often the list of
that order holds rectangle i (round 6: the lists are multisets - the reference repeats a rectangle that two boundary components vote for), `_member` = count > 0.
The frames are the ones tools/stream_mismatch.py (GPU box) reported.  Only runs where /root/reference exists (oracle/_ref)."""
import ctypes
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rectdetect_amd as ra  # noqa: E402
from rectdetect_amd import synth  # noqa: E402
from tests import helpers  # noqa: E402
from tools.make_golden_orders import FILTER, rect_key  # noqa: E402
from tools.make_golden_orders import ORDERS as STILL_ORDERS  # noqa: E402

# the 26 sequential orders of the busy stills + the CONCURRENT execution of labelMergeMain (no work-item sees another one's update
# within a launch: rdcl_set_order group_order 5; despeckle2, which stores plainly, in raster order) + more scrambled walks
ORDERS = STILL_ORDERS + [(0, 0, 5, 0), (64, 4, 3, 7), (16, 16, 3, 9), (128, 2, 3, 3), (32, 32, 3, 1), (4, 4, 3, 2)]

# (round 2: the frames on which the HIP path of that round differed; round 3 - region merge with concurrent work-items, exact absorption - adds 161 / 70, 76, 90)
FRAMES = {"stream_1280x720_s1_300": [32, 161, 162], "stream_1920x1080_s0_100": [42, 55, 57, 70, 76, 90, 98], "stream_3840x2160_s4_16": [3, 7],
          "stream_1920x1080_s7_100": [0, 5, 63, 64, 71, 72],      # (the held-out stream of round 3: tools/stream_mismatch.py)
          # round 5's two held-out streams (generated after the round's last kernel change; 64 frames in flight)
          "stream_1920x1080_s11_200": [0, 2, 12, 28, 31, 49, 112, 119, 155], "stream_1920x1080_s12_200": [14, 25, 26, 30, 39],
          "stream_1280x720_s13_300": [112, 147, 158, 267, 277, 297],
          # round 6's held-out streams: the frames whose multiset of rectangles under the concurrent, settled merge differs from the raster order's and whose merge had not settled
          # within the reference's 8 launches (tests/golden/*_settled.npz against the raster-order goldens)
          "stream_3840x2160_s4_32": [17, 19, 23, 25, 27, 30], "stream_1920x1080_s21_300": [139, 185],
          "stream_1920x1080_s22_300": [4, 74, 118, 159, 179, 195, 229, 237, 285],
          "stream_1920x1080_s23_300": [99, 116, 119, 130, 244, 269], "stream_1920x1080_s24_300": [8, 27, 43, 208],
          # (configs[3] at SURVEY.md 8(d)'s length; its first 32 frames are stream_3840x2160_s4_32: frames 17 .. 30 are copied from there)
          "stream_3840x2160_s4_100": [33, 35, 37, 38, 41, 47, 48, 50, 51, 53, 56, 59, 67, 72, 75, 78, 81, 85, 87, 88, 89, 92, 95, 97, 98]}
PARALLEL = int(os.environ.get("RD_ORDERS_PARALLEL", "8"))


def one_stream(name, frames, out):
    g = np.load(os.path.join(helpers.GOLDEN, name + ".npz"), allow_pickle=False)
    iw, ih, tan, seed = int(g["iw"]), int(g["ih"]), float(g["tan_aov"]), int(g["seed"])
    R = helpers.ref()
    R.rdcl_set_order.argtypes = [ctypes.c_char_p] + [ctypes.c_int] * 4
    r = helpers.RefRect(iw, ih)
    with tempfile.TemporaryDirectory() as td:
        for t in range(max(frames) + 1):
            img = synth.frame(seed, iw, ih, t)
            if t in frames:
                want_segments = g[f"f{t}_segments"]      # (read before forking: the copies share the archive's file offset)
                pending = list(range(len(ORDERS)))
                running = {}
                while pending or running:
                    while pending and len(running) < PARALLEL:
                        oi = pending.pop(0)
                        pid = os.fork()
                        if pid == 0:            # the copy: this frame under order oi, then gone
                            R.rdcl_set_order(FILTER, *ORDERS[oi])
                            rects, snaps = r.execute_once(img, tan, snapshots=["lslist"])
                            n = int(snaps["lslist"][0])
                            ok = helpers.segments_equal(snaps["lslist"][: 14 * (n + 1)].view(ra.LS_DTYPE), want_segments)
                            np.savez(os.path.join(td, "%d_%d.npz" % (t, oi)), rects=rects, segments_ok=ok)
                            os._exit(0)
                        running[pid] = oi
                    pid, status = os.wait()
                    assert status == 0, (name, t, running[pid], status)
                    del running[pid]
                lists = []
                for oi in range(len(ORDERS)):
                    with np.load(os.path.join(td, "%d_%d.npz" % (t, oi))) as z:
                        assert bool(z["segments_ok"]), "segments must not depend on the order of the region kernels"
                        lists.append(z["rects"])
                assert helpers.rects_equal(lists[0], g[f"f{t}_rects"]), "order 0 is the raster order of the golden stream"
                union, index = [], {}
                for l in lists:
                    for q in l:
                        if rect_key(q) not in index:
                            index[rect_key(q)] = len(union)
                            union.append(q)
                # count[order, i] = how many times the list of that order holds rectangle i (the reference lists a rectangle once per boundary component that votes for
                # it: exact duplicates are part of its output, and whether two components are one depends on the order like everything else); member = count > 0
                count = np.zeros((len(ORDERS), len(union)), np.uint8)
                for oi, l in enumerate(lists):
                    for q in l:
                        count[oi, index[rect_key(q)]] += 1
                member = (count > 0).astype(np.uint8)
                out[f"{name}_f{t}_union"] = np.array(union, dtype=ra.RECT_DTYPE) if union else np.zeros(0, ra.RECT_DTYPE)
                out[f"{name}_f{t}_member"] = member
                out[f"{name}_f{t}_count"] = count
                print(name, "frame", t, "rectangles per order", count.sum(1).tolist(), "distinct", len(union), "in every order", int(member.all(0).sum()),
                      "listed more than once under some order", int((count.max(0) > 1).sum()), "with a multiplicity that depends on the order", int((count.max(0) != count.min(0)).sum()), flush=True)
            rects, _ = r.execute_once(img, tan)          # the stream itself goes on in raster order
            assert helpers.rects_equal(rects, g[f"f{t}_rects"]), (name, t)
    r.close()


def main():
    path = os.environ.get("RD_ORDERS_OUT") or os.path.join(helpers.GOLDEN, "stream_orders.npz")      # (RD_ORDERS_OUT: streams run side by side, merged afterwards)
    out = {"orders": np.array(ORDERS, np.int32), "filter": FILTER.decode()}
    if os.path.exists(path) and len(sys.argv) > 1:           # (one stream at a time: keep what is there)
        with np.load(path) as z:
            out.update({k: z[k] for k in z.files})
    for name in sys.argv[1:] or sorted(FRAMES):
        one_stream(name, FRAMES[name], out)
    np.savez_compressed(path, **out)


if __name__ == "__main__":
    main()
