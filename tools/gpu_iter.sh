#!/bin/bash
# one optimisation-loop iteration on the GPU box: parity tests, bench points, one kernel trace, one SQ counter pass
# usage (from the repo root, on the box): bash tools/gpu_iter.sh <tag>
tag=${1:-x}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -15 > gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
for s in 1 4 8; do
  timeout 300 python bench.py --steps 3 --warmup 1 --slots $s --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_s$s.log | cut -c1-100
done
export TMPDIR=/tmp
R=$PWD
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof$tag -o r$tag -- python $R/bench.py --steps 1 --warmup 1 --slots 1 --frames-per-step 8 --no-cpu-baseline > $R/gpurun_out/prof$tag.log 2>&1
RD_NO_GRAPH=1 timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/sq$tag -o sq -- python $R/bench.py --steps 1 --warmup 1 --slots 1 --frames-per-step 4 --no-cpu-baseline > $R/gpurun_out/sq$tag.log 2>&1
cd $R
