#!/bin/bash
# on the GPU box: frames/s of tuning builds (tools/variants.sh), interleaved with the default build, two rounds
for rep in 1 2; do
for v in default "$@"; do
  if [ $v = default ]; then unset RD_LIB_PATH; else export RD_LIB_PATH=$PWD/rectdetect_amd/variants/lib$v.so; fi
  python bench.py --steps 12 --warmup 3 --frames-per-step 128 --no-cpu-baseline --no-configs --no-verify 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$v', j['value'])"
done; done
