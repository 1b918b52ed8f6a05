#!/bin/bash
# one optimisation-loop iteration on the GPU box: parity tests, two bench points, one kernel trace
# usage (from the repo root, on the box): bash tools/gpu_iter.sh <tag>
tag=${1:-x}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -15 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
timeout 300 python bench.py --steps 3 --warmup 1 --slots 1 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_s1.log | cut -c1-120
timeout 300 python bench.py --steps 3 --warmup 1 --slots 8 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_s8.log | cut -c1-120
export TMPDIR=/tmp
R=$PWD
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof$tag -o r$tag -- python $R/bench.py --steps 1 --warmup 1 --slots 1 --frames-per-step 8 --no-cpu-baseline > $R/gpurun_out/prof$tag.log 2>&1
cd $R
find gpurun_out/prof$tag -name "*.db" | head -2
