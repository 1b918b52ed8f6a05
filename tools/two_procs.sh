#!/bin/bash
# diagnostics on the GPU box: two independent bench processes on the same GPU at the same time (do more hardware queues from another process add throughput?)
run() { python bench.py --steps 12 --warmup 3 --frames-per-step 128 --slots ${1:-16} --no-cpu-baseline --no-configs --no-verify 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('proc', j['value'], 'frames/s', j['ms_per_step'])"; }
echo "one process:"; run 16
echo "two processes:"; run 16 & run 16 & wait
echo "two processes, 8 slots each:"; run 8 & run 8 & wait
