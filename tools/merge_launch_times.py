#!/usr/bin/env python3
"""average duration of every launch of the region merge (k_region_round) from a rocprofv3 --kernel-trace results.db: usage merge_launch_times.py <db>"""
import sqlite3, collections, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, stream_id from kernels order by start"))
per = collections.defaultdict(list); cur = {}
for name, st, en, sid in rows:
    if "k_region_init" in name: cur[sid] = []
    elif "k_region_round" in name and sid in cur: cur[sid].append((en - st) / 1000.0)
    elif "k_region_size" in name and sid in cur:
        for i, d in enumerate(cur[sid]): per[i].append(d)
        del cur[sid]
print(" ".join("%d:%.1f" % (i + 1, sum(per[i]) / len(per[i])) for i in sorted(per)))
