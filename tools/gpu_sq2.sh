#!/bin/bash
# SQ counter pass over the benchmarked configuration (group launches, no graphs) + per-kernel summary: bash tools/gpu_sq2.sh <tag> [counters...]
tag=${1:-x}; shift
C=${@:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES}
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp
RD_NO_GRAPH=1 timeout 900 rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/sq$tag -o sq -- python $R/bench.py --steps 1 --warmup 1 --frames-per-step 64 --no-cpu-baseline --no-verify --no-configs > $R/gpurun_out/sq$tag.log 2>&1
cd $R
python tools/pmc_summary.py $(find gpurun_out/sq$tag -name "*results.db" | head -1) 128 45 > gpurun_out/sq_$tag.txt
cat gpurun_out/sq_$tag.txt | head -40
