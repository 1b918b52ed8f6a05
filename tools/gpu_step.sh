#!/bin/bash
# one development step on the GPU box: the GPU tests (optionally -k <expr>), A/B bench points from a file of gpu_ab.sh lines (interleaved
# twice), a kernel trace of the default configuration distilled to microseconds per frame.
# usage (on the box): bash tools/gpu_step.sh <tag> <ab file or -> [pytest -k expression]
tag=${1:-x}; ab=${2:--}
mkdir -p gpurun_out
if [ -n "$3" ]; then K=(-k "$3"); else K=(); fi
timeout 1500 python -m pytest tests -m gpu -x -q "${K[@]}" 2>&1 | tail -25 > gpurun_out/pytest_$tag.log
tail -4 gpurun_out/pytest_$tag.log
if [ "$ab" != "-" ]; then for i in 1 2; do bash tools/gpu_ab.sh < $ab; done; fi
export TMPDIR=/tmp
R=$PWD
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof$tag -o r$tag -- python $R/bench.py --steps 3 --warmup 1 --frames-per-step 64 --no-cpu-baseline --no-verify --no-configs > $R/gpurun_out/prof$tag.log 2>&1
cd $R
python tools/prof_summary.py $(find gpurun_out/prof$tag -name "*results.db" | head -1) 256 > gpurun_out/kstats_$tag.txt 2>&1
head -45 gpurun_out/kstats_$tag.txt
