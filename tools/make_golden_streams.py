#!/usr/bin/env python3
"""Longer streams from THE REFERENCE (oracle/_ref), consecutive frames of one detector instance, so that the state the
reference carries from frame to frame takes part well beyond the 2-3 frames of the other fixtures:
  stream_1920x1080_s0.npz  16 frames of the benchmark workload (BASELINE.json configs[4]: 1920x1080, stream seed 0)
  stream_1280x720_s1.npz   30 frames of configs[2] (1280x720, AOV 72 like vidrect's default)
  stream_3840x2160_s4.npz  3 frames of configs[3] (3840x2160)
and the configurations at (or near) their full length (rect_t lists and segment lists only, a few hundred KB each):
  stream_1280x720_s1_300.npz    all 300 frames of configs[2]
  stream_1920x1080_s0_100.npz   100 frames of the benchmark stream
  stream_3840x2160_s4_16.npz    16 frames of configs[3]
  stream_1920x1080_s7_100.npz   100 frames of another 1920x1080 stream (seed 7): held out
Per frame: the rect_t list and the line-segment list.  Only runs where /root/reference exists."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rectdetect_amd as ra  # noqa: E402
from rectdetect_amd import synth  # noqa: E402
from tests import helpers  # noqa: E402

CASES = {"stream_1920x1080_s0": (1920, 1080, 0, 16, 36.0), "stream_1280x720_s1": (1280, 720, 1, 30, 36.0),
         "stream_3840x2160_s4": (3840, 2160, 4, 3, 36.0),      # BASELINE.json configs[3]
         "stream_1280x720_s1_300": (1280, 720, 1, 300, 36.0), "stream_1920x1080_s0_100": (1920, 1080, 0, 100, 36.0),
         "stream_3840x2160_s4_16": (3840, 2160, 4, 16, 36.0),
         # round 3: a second 1920x1080 stream (another seed) that no design decision was made on - a held-out check of the region stages
         "stream_1920x1080_s7_100": (1920, 1080, 7, 100, 36.0),
         # round 5: two more 1920x1080 streams of other seeds, 200 frames each, generated after the last kernel change of the round - held out as well
         "stream_1920x1080_s11_200": (1920, 1080, 11, 200, 36.0), "stream_1920x1080_s12_200": (1920, 1080, 12, 200, 36.0),
         "stream_1280x720_s13_300": (1280, 720, 13, 300, 36.0),      # (and a second 1280x720 stream at configs[2]'s length, another seed)
         # round 6: SURVEY.md 8(d)'s depth - two more 1920x1080 streams at 300 frames (other seeds) and configs[3] at 32 frames -, generated after the round's last kernel
         # change: held out like the streams of round 5
         "stream_1920x1080_s21_300": (1920, 1080, 21, 300, 36.0), "stream_1920x1080_s22_300": (1920, 1080, 22, 300, 36.0),
         "stream_1920x1080_s23_300": (1920, 1080, 23, 300, 36.0), "stream_1920x1080_s24_300": (1920, 1080, 24, 300, 36.0),
         "stream_3840x2160_s4_32": (3840, 2160, 4, 32, 36.0), "stream_3840x2160_s4_100": (3840, 2160, 4, 100, 36.0)}      # (100: SURVEY.md 8(d)'s t = 0..99)
DEFAULT = ["stream_1920x1080_s0", "stream_1280x720_s1", "stream_3840x2160_s4"]      # (the long ones by name: 6-10 minutes each)


def main():
    for name in sys.argv[1:] or DEFAULT:
        iw, ih, seed, nframes, half_aov = CASES[name]
        tan = float(np.tan(half_aov / 180.0 * np.pi))
        r = helpers.RefRect(iw, ih)
        out = {"iw": iw, "ih": ih, "seed": synth.SEED0 + seed, "nframes": nframes, "tan_aov": tan}
        for t in range(nframes):
            rects, snaps = r.execute_once(synth.frame(synth.SEED0 + seed, iw, ih, t), tan, snapshots=["lslist"])
            n = int(snaps["lslist"][0])
            out[f"f{t}_rects"] = rects
            out[f"f{t}_segments"] = snaps["lslist"][: 14 * (n + 1)].view(ra.LS_DTYPE)
            print(name, "frame", t, "rects", len(rects), "segments", n, flush=True)
            if os.environ.get("RD_GOLDEN_PREFIXES") and (t + 1) % 50 == 0 and t + 1 < nframes:      # (long runs: the stream so far, as a fixture of its own length, in case the run is cut short)
                np.savez_compressed(os.path.join(os.environ["RD_GOLDEN_PREFIXES"], "%s_first%d.npz" % (name, t + 1)), **dict(out, nframes=t + 1))
        r.close()
        np.savez_compressed(os.path.join(helpers.GOLDEN, name + ".npz"), **out)


if __name__ == "__main__":
    main()
