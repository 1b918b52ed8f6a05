#!/bin/bash
# one optimisation-loop iteration on the GPU box: parity tests, bench points, one kernel trace
# usage (from the repo root, on the box): bash tools/gpu_iter.sh <tag> [pytest -k expression]
tag=${1:-x}
mkdir -p gpurun_out
if [ -n "$2" ]; then K=(-k "$2"); else K=(); fi
timeout 1200 python -m pytest tests -m gpu -x -q "${K[@]}" 2>&1 | tail -15 > gpurun_out/pytest_$tag.log
tail -6 gpurun_out/pytest_$tag.log
for s in 1 4 8; do
  timeout 300 python bench.py --steps 3 --warmup 1 --slots $s --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_${tag}_s$s.log | cut -c1-120
done
export TMPDIR=/tmp
R=$PWD
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof$tag -o r$tag -- python $R/bench.py --steps 1 --warmup 1 --slots 1 --frames-per-step 8 --no-cpu-baseline --no-verify > $R/gpurun_out/prof$tag.log 2>&1
cd $R
python tools/prof_summary.py $(find gpurun_out/prof$tag -name "*results.db" | head -1) 16 > gpurun_out/kstats_$tag.txt 2>&1
head -30 gpurun_out/kstats_$tag.txt
