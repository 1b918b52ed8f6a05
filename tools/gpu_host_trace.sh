#!/bin/bash
# on the GPU box: where does a host frame's upload go?  kernel + memory-copy trace of a short --host-frames run (pinned and pageable), copy statistics and the kernels'
# totals beside a resident run's.  bash tools/gpu_host_trace.sh
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
for mode in resident pinned pageable; do
  extra=""; [ $mode != resident ] && extra="--host-frames --host-mode $mode"
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $R/gpurun_out/ht_$mode -o t -- python $R/bench.py --steps 4 --warmup 2 --frames-per-step 128 --no-cpu-baseline --no-verify $extra > $R/gpurun_out/ht_$mode.log 2>&1
  cd $R
  echo "== $mode: $(tail -1 gpurun_out/ht_$mode.log | python -c 'import json,sys; j=json.loads(sys.stdin.readline()); print(j["value"], "frames/s")' 2>/dev/null)"
  f=$(find gpurun_out/ht_$mode -name "*memory_copy_stats.csv" | head -1); [ -n "$f" ] && cat $f | head -8
  f=$(find gpurun_out/ht_$mode -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 $f | cut -c1-160
  f=$(find gpurun_out/ht_$mode -name "*memory_copy_trace.csv" | head -1); [ -n "$f" ] && python - "$f" <<PY
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
big=[r for r in rows if int(r.get("End_Timestamp",0))-int(r.get("Start_Timestamp",0))>0]
print("copies", len(rows), "columns", list(rows[0].keys()) if rows else None)
import collections
by=collections.defaultdict(list)
for r in rows:
    by[(r.get("Direction"), r.get("Source_Agent_Id"), r.get("Destination_Agent_Id"))].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in by.items():
    v.sort(); print(k, "n", len(v), "median us", v[len(v)//2], "p90", v[len(v)*9//10], "max", v[-1])
PY
  rm -rf gpurun_out/ht_$mode/*/*.db 2>/dev/null
done
