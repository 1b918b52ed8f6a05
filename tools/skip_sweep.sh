#!/bin/bash
# diagnostics on the GPU box: frames/s with stages left out (RD_DIAG_SKIP bits: 1 = two blur pairs instead of ten, 2 = two merge launches only, 4 = no polyline stage) - wrong results, timing only
for sk in ${@:-0 1 2 3 4}; do
  RD_DIAG_SKIP=$sk python bench.py --steps 10 --warmup 3 --frames-per-step 64 --no-cpu-baseline --no-configs --no-verify 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('RD_DIAG_SKIP=$sk', j['value'], 'frames/s')"
done
