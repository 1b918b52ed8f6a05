#!/bin/bash
# frames/s by frames in flight (sparse stages in batches of 4 from 12 slots on, of 8 from 24 on; RD_BATCH overrides)
for sl in ${@:-16 24 32}; do
  python bench.py --steps 10 --warmup 3 --frames-per-step 128 --slots $sl --no-cpu-baseline --no-configs --no-verify 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('slots $sl RD_BATCH=$RD_BATCH', j['value'], 'frames/s')"
done
