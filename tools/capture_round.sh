#!/bin/bash
# The measurement set of a state of the code, end to end (run HERE, in the build container):  bash tools/capture_round.sh <tag> <prefix, e.g. r03_a>
# refuses to run on a dirty tree, records HEAD, runs tools/gpu_capture.sh on the GPU box and distils the result into profiles/<prefix>_*
# (tools/distill_capture.py fails if HEAD moved meanwhile: profiles/ must describe the commit that is benchmarked)
set -e
tag=$1; pre=$2
cd "$(dirname "$0")/.."
if [ -n "$(git status --porcelain -- rectdetect_amd bench.py tools/gpu_capture.sh tools/gpu_pmc.sh)" ]; then echo "capture_round: commit first"; exit 1; fi
git rev-parse --short HEAD > tools/.capture_commit
/usr/local/graft/bin/gpurun --timeout 2400 -- "bash tools/gpu_capture.sh $tag > gpurun_out/capture_$tag.log 2>&1; tail -5 gpurun_out/capture_$tag.log"
python tools/distill_capture.py $tag $pre
