#!/bin/bash
# SQ counter pass (one frame in flight, no graphs) + per-kernel summary: bash tools/gpu_sq.sh <tag>
tag=${1:-x}
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp
RD_NO_GRAPH=1 timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/sq$tag -o sq -- python $R/bench.py --steps 1 --warmup 1 --slots 1 --frames-per-step 4 --no-cpu-baseline --no-verify > $R/gpurun_out/sq$tag.log 2>&1
cd $R
python tools/pmc_summary.py $(find gpurun_out/sq$tag -name "*results.db" | head -1) 8 45 > gpurun_out/sq_$tag.txt
cut -c1-45,100-135 gpurun_out/sq_$tag.txt | head -50
