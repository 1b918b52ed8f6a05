#!/bin/bash
# on the GPU box: kernel trace of the default bench run and the device-occupancy summary of its middle: bash tools/gpu_timeline.sh [bench args]
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
rm -rf /tmp/tl
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d /tmp/tl -o k -- python $R/bench.py --steps 2 --warmup 1 --frames-per-step 128 --no-cpu-baseline --no-verify --no-configs "$@" > /tmp/tl.log 2>&1)
tail -1 /tmp/tl.log | cut -c1-120
db=$(find /tmp/tl -name "*results.db" | head -1); python tools/prof_timeline.py $db 0.5 | tee gpurun_out/timeline.txt; python tools/prof_gaps.py $db 40 10 | tee gpurun_out/gaps.txt
