#!/bin/bash
# on the GPU box: which hardware queues the headline run's streams land on and where they idle (kernel trace of a short run)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tq && mkdir -p /tmp/tq
rocprofv3 --kernel-trace -d /tmp/tq -o tq -- python $R/bench.py --steps 3 --warmup 1 --frames-per-step 256 --no-cpu-baseline --no-verify --no-configs 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j["value"])'
python $R/tools/trace_queues.py $(find /tmp/tq -name "*.db" | head -1) | tail -n +2
python $R/tools/queue_gaps.py $(find /tmp/tq -name "*.db" | head -1) 10
