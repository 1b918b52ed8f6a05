#!/bin/bash
# interleaved variant bench with the longer default stream
for rep in 1 2 3; do
for v in default "$@"; do
  if [ $v = default ]; then unset RD_LIB_PATH; else export RD_LIB_PATH=$PWD/rectdetect_amd/variants/lib$v.so; fi
  python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-configs --no-verify 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$v', j['value'])"
done; done
