#!/bin/bash
# on the GPU box: per-kernel time of tuning builds under a kernel trace: bash tools/gpu_kvar.sh <kernel name pattern> <variant|default> ...
pat=$1; shift
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
for v in "$@"; do
  if [ $v = default ]; then unset RD_LIB_PATH; else export RD_LIB_PATH=$R/rectdetect_amd/variants/lib$v.so; fi
  rm -rf /tmp/kv_$v
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kv_$v -o k -- python $R/bench.py --steps 2 --warmup 1 --frames-per-step 64 --no-cpu-baseline --no-verify --no-configs > /tmp/kv_$v.log 2>&1)
  fps=$(tail -1 /tmp/kv_$v.log | python -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])' 2>/dev/null)
  echo "== $v (traced run: $fps frames/s)" | tee -a gpurun_out/kvar.log
  python tools/prof_summary.py $(find /tmp/kv_$v -name "*results.db" | head -1) 192 | grep -E "^#|$pat" | tee -a gpurun_out/kvar.log
done
