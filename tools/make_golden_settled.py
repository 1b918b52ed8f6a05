#!/usr/bin/env python3
"""What THE REFERENCE ITSELF returns when its region merge runs the way the HIP path runs it: `rect:labelMergeMain` with the work-items of a launch
concurrent (rdcl_set_order group_order 5: a legal execution of the kernel, order 26 of tests/golden/stream_orders.npz) and launched until a launch changes
nothing instead of the 8 times of oclrect.c:325-331 (rdcl_set_repeat: the serial stand-in runs the very same launch again).  Everything else - the other 219
launches, the read-backs, executeCPUTask - is the reference's own compiled code, unchanged (oracle/_ref).

-> tests/golden/<stream>_settled.npz, per frame t:
     f<t>_rects            the rect_t list, in the reference's list order
     launches[t]           launches of the merge kernel up to and including the first that changed nothing (8 = settled within the reference's own 8)
     settled_after_8[t]    the reference's 8th launch (concurrent work-items) left nothing for a 9th to change
     changed_9th[t]        label words the 9th launch changes (0 where settled)
   The segment lists do not depend on any of this (asserted against the raster-order golden of the same stream).
Only runs where /root/reference exists.  usage: python tools/make_golden_settled.py <stream> [...]   (11 minutes per 200 frames of 1920x1080)"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rectdetect_amd as ra  # noqa: E402
from rectdetect_amd import synth  # noqa: E402
from tests import helpers  # noqa: E402

MERGE = b"rect:labelMergeMain"
EXTRA = 120      # (the HIP path gives up at 64 launches; no frame of the fixtures comes near)


def one_stream(name):
    g = np.load(os.path.join(helpers.GOLDEN, name + ".npz"), allow_pickle=False)
    iw, ih, tan, seed, nframes = int(g["iw"]), int(g["ih"]), float(g["tan_aov"]), int(g["seed"]), int(g["nframes"])
    R = helpers.ref()
    R.rdcl_set_order.argtypes = [ctypes.c_char_p] + [ctypes.c_int] * 4
    R.rdcl_set_repeat.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    r = helpers.RefRect(iw, ih)
    out = {"iw": iw, "ih": ih, "seed": seed, "nframes": nframes, "tan_aov": tan}
    launches, settled8, changed9 = [], [], []
    for t in range(nframes):
        img = synth.frame(seed, iw, ih, t)
        R.rdcl_set_order(MERGE, 0, 0, 5, 0)
        R.rdcl_set_repeat(MERGE, 7, EXTRA)
        rects, snaps = r.execute_once(img, tan, snapshots=["lslist"])
        ch = [R.rdcl_repeat_changed(i) for i in range(EXTRA)]
        R.rdcl_set_repeat(b"", -1, 0)
        R.rdcl_set_order(b"", 0, 0, 0, 0)
        assert 0 in ch, (name, t, "the merge did not settle within %d launches" % (8 + EXTRA))
        n = int(snaps["lslist"][0])
        assert helpers.segments_equal(snaps["lslist"][: 14 * (n + 1)].view(ra.LS_DTYPE), g[f"f{t}_segments"]), (name, t, "segments must not depend on the region merge")
        out[f"f{t}_rects"] = rects
        launches.append(8 + ch.index(0) + 1)      # the launch that changed nothing is counted, as the HIP path's `need` counts it
        settled8.append(ch[0] == 0)
        changed9.append(ch[0])
        print(name, "frame", t, "rects", len(rects), "raster-order golden", len(g[f"f{t}_rects"]), "same list:", helpers.rects_equal(rects, g[f"f{t}_rects"]),
              "| launches", launches[-1], "changed by the 9th", ch[0], flush=True)
    r.close()
    out["launches"] = np.array(launches, np.int32)
    out["settled_after_8"] = np.array(settled8, np.uint8)
    out["changed_9th"] = np.array(changed9, np.int32)
    np.savez_compressed(os.path.join(helpers.GOLDEN, name + "_settled.npz"), **out)


if __name__ == "__main__":
    for name in sys.argv[1:]:
        one_stream(name)
