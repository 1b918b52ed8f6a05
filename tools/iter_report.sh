#!/bin/bash
# local: summarise what gpu_iter.sh <tag> brought back
tag=$1
python tools/prof_summary.py gpurun_out/prof$tag/r${tag}_results.db 16 2>/dev/null | head -${2:-12}
python - <<PY
import sqlite3, collections
db = sqlite3.connect('gpurun_out/sq$tag/sq_results.db')
acc = collections.defaultdict(float)
for k, v in db.execute("select kernel_name, value from counters_collection where counter_name='SQ_WAVE_CYCLES'"):
    acc[k.replace('(anonymous namespace)::','').split('(')[0].split('::')[-1].replace('void ','')] += v / 8
tot = sum(acc.values())
print("SQ_WAVE_CYCLES per frame: %.0fM" % (tot / 1e6))
print("  ".join("%s %.0fM" % (k[:22], v / 1e6) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:${3:-18}]))
PY
