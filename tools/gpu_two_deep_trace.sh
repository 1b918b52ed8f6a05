#!/bin/bash
# on the GPU box: kernel trace of the two-deep call sequence; a window of it by queue -> gpurun_out/two_deep_window.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/td && mkdir -p /tmp/td
rocprofv3 --kernel-trace -d /tmp/td -o td -- python $R/tools/two_deep.py 200 2>&1 | grep "two deep"
python $R/tools/window_timeline.py $(find /tmp/td -name "*.db" | head -1) 1800 > $R/gpurun_out/two_deep_window.txt 2>&1
python $R/tools/trace_queues.py $(find /tmp/td -name "*.db" | head -1)
