#!/bin/bash
# A/B bench points on the GPU box: each line of stdin is "label [ENV=..]... [-- bench args]"; prints frames/s and the average per-frame
# device interval of `bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-verify <args>`
mkdir -p gpurun_out
while read -r label rest; do
  [ -z "$label" ] && continue
  envs="${rest%%--*}"; args=""
  case "$rest" in *--*) args="${rest#*--}";; esac
  v=$(env X=1 $envs timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-verify $args 2>/dev/null | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["roofline"].get("frame_device_us_avg"), "enqueue us/frame", j["roofline"]["host_enqueue_us_avg"])')
  echo "$label: $v" | tee -a gpurun_out/ab.log
done
