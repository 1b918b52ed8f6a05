#!/bin/bash
# on the GPU box: kernel trace of single frames through the reference API shape (one frame in flight), then one frame's time line.  bash tools/gpu_timeline.sh [lib]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tl && mkdir -p /tmp/tl
[ -n "$1" ] && export RD_LIB_PATH=$R/rectdetect_amd/variants/lib$1.so
rocprofv3 --kernel-trace -d /tmp/tl -o tl -- python $R/tools/lat_api.py > $R/gpurun_out/timeline_lat.txt 2>&1
DB=$(find /tmp/tl -name "*.db" | head -1)
python $R/tools/frame_timeline.py $DB 3 > $R/gpurun_out/timeline.txt 2>&1
tail -3 $R/gpurun_out/timeline_lat.txt; head -3 $R/gpurun_out/timeline.txt
