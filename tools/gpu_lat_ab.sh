#!/bin/bash
# on the GPU box: tools/lat_api.py (host post-process only) under the settings of stdin-file lines ("label [ENV=..]...").  bash tools/gpu_lat_ab.sh <file> [N=1]
mkdir -p gpurun_out
export LAT_HOST_POST_ONLY=1
for i in $(seq 1 ${2:-1}); do
  while read -r label rest; do
    [ -z "$label" ] && continue
    echo "$label: $(env X=1 $rest timeout 120 python tools/lat_api.py 2>&1 | tail -1)" | tee -a gpurun_out/lat_ab.log
  done < $1
done
