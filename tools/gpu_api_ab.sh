#!/bin/bash
# on the GPU box: the reference-API rows under the settings of stdin's lines ("label [ENV=..]..."), interleaved N times.  bash tools/gpu_api_ab.sh <file> [N=2]
mkdir -p gpurun_out
for i in $(seq 1 ${2:-2}); do
  while read -r label rest; do
    [ -z "$label" ] && continue
    echo "$label: $(env X=1 $rest timeout 120 python tools/api_rate.py 2>&1 | tail -1)" | tee -a gpurun_out/api_ab.log
  done < $1
done
