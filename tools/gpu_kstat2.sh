#!/bin/bash
# on the GPU box: per-kernel microseconds per frame of the default bench configuration for tuning builds: bash tools/gpu_kstat2.sh <variant> <variant> ... [-- grep pattern]
export TMPDIR=/tmp
R=$PWD
pat="total"
vars=()
for a in "$@"; do if [ "$a" = "--" ]; then shift; pat="total|$*"; break; fi; vars+=("$a"); shift; done
for v in "${vars[@]}"; do
  lib=""; [ "$v" != "default" ] && lib="RD_LIB_PATH=$R/rectdetect_amd/variants/lib$v.so"
  cd /tmp && env X=1 $lib timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ks_$v -o t -- python $R/bench.py --steps 3 --warmup 1 --frames-per-step 64 --no-cpu-baseline --no-verify --no-configs > $R/gpurun_out/ks_$v.log 2>&1
  cd $R
  echo "== $v"; python tools/prof_summary.py $(find gpurun_out/ks_$v -name "*results.db" | head -1) 256 | grep -E "$pat"
  rm -rf gpurun_out/ks_$v
done
