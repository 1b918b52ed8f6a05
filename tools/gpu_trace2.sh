#!/bin/bash
# kernel-trace statistics of the default bench configuration under two environments: bash tools/gpu_trace2.sh "ENV_A=.." "ENV_B=.."
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
i=0
for e in "$@"; do
  i=$((i+1))
  cd /tmp && env $e timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/tr$i -o t -- python $R/bench.py --steps 2 --warmup 1 --frames-per-step 32 --no-cpu-baseline --no-verify > $R/gpurun_out/tr$i.log 2>&1
  cd $R
  echo "== $e"; python tools/prof_summary.py $(find gpurun_out/tr$i -name "*results.db" | head -1) 96 | head -${LINES_OUT:-45}
done
