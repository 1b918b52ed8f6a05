#!/bin/bash
# on the GPU box: which hardware queues the headline run's streams land on (kernel trace of a short run)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tq && mkdir -p /tmp/tq
rocprofv3 --kernel-trace -d /tmp/tq -o tq -- python $R/bench.py --steps 2 --warmup 1 --frames-per-step 128 --no-cpu-baseline --no-verify --no-configs > /dev/null 2>&1
python $R/tools/trace_queues.py $(find /tmp/tq -name "*.db" | head -1)
