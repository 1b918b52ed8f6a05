#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 --pmc counters from a results.db (table counters_collection).
usage: pmc_summary.py <results.db> <frames> [top]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); frames = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rows = db.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
names = []
for k, c, v in rows:
    k = k.replace('(anonymous namespace)::', '').split('(')[0].split('::')[-1].replace('void ', '')
    acc[k][c] += v
    if c not in names: names.append(c)
first = names[0]
for k, c, v in rows:
    if c == first: calls[k.replace('(anonymous namespace)::', '').split('(')[0].split('::')[-1].replace('void ', '')] += 1
tot = collections.defaultdict(float)
for k in acc:
    for c in names: tot[c] += acc[k][c]
print("# per frame (%g frames); counters: %s" % (frames, ", ".join(names)))
print("%-36s %8s " % ("kernel", "calls/fr") + " ".join("%16s" % c[-16:] for c in names))
for k in sorted(acc, key=lambda k: -acc[k][first])[:top]:
    print("%-36s %8.1f " % (k[:36], calls[k] / frames) + " ".join("%16.0f" % (acc[k][c] / frames) for c in names))
print("%-36s %8s " % ("TOTAL", "") + " ".join("%16.0f" % (tot[c] / frames) for c in names))
