#!/bin/bash
# on the GPU box: the two reference-API configurations of the bench line (two-deep enqueue / poll, executeOnce), alternating environments: bash tools/gpu_refapi.sh "ENV=.." "ENV=.." ...
for e in "$@"; do
  env X=1 $e python - <<'PY'
import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np
import bench, rectdetect_amd as ra
from rectdetect_amd import synth
L = ra.lib()
frames = []
for t in range(64):
    a = np.zeros((1080, 1920, 3), np.uint8); L.rd_synth_frame(a.ctypes.data, 1920, 1080, 1920 * 3, synth.SEED0, t, 1); frames.append(a)
print(os.environ.get("RD_UNUSED", "-"), [(c["workload"][:24], c["value"], c.get("latency_ms_median")) for c in bench.reference_api_config(ra, frames, 0)])
PY
done
