import sys, json, numpy as np, os
sys.path.insert(0, ".")
import bench, rectdetect_amd as ra
from rectdetect_amd import synth
L = ra.lib()
frames = []
for t in range(16):
    a = np.zeros((1080, 1920, 3), np.uint8)
    L.rd_synth_frame(a.ctypes.data, 1920, 1080, 1920 * 3, synth.SEED0, t, 1)
    frames.append(a)
for r in bench.reference_api_config(ra, frames, 0):
    print(os.environ.get("TAG",""), r["workload"][:75], r["value"], r.get("latency_ms_median"))
