#!/usr/bin/env python3
"""Idle time of each hardware queue in a kernel trace (second half of the trace): python tools/queue_gaps.py <results.db> [min gap us = 10]
-> per queue: span, busy, idle in gaps >= min; the kernels most often found after / before such a gap"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
ming = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table' or type='view'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(db.execute(f"select d.start, d.end, s.kernel_name, d.queue_id from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
rows = rows[len(rows) // 2:]
def short(n):
    n = n.split("(")[0]
    return n[n.find("k_"):][:30] if "k_" in n else n[:30]
byq = collections.defaultdict(list)
for r in rows: byq[r[3]].append(r)
for q, rs in sorted(byq.items()):
    if len(rs) < 50: continue
    span = (rs[-1][1] - rs[0][0]) / 1e3
    busy = sum(r[1] - r[0] for r in rs) / 1e3
    after, before, idle, big = collections.Counter(), collections.Counter(), 0.0, []
    for a, b in zip(rs, rs[1:]):
        g = (b[0] - a[1]) / 1e3
        if g >= ming:
            idle += g; after[short(b[2])] += g; before[short(a[2])] += g; big.append(g)
    big.sort()
    print("queue %s: %d launches, span %.0f us, busy %.0f (%.1f %%), idle in gaps >= %.0f us: %.0f us (%.1f %%), %d gaps, median %.0f, max %.0f" % (q, len(rs), span, busy, 100 * busy / span, ming, idle, 100 * idle / span, len(big), big[len(big) // 2] if big else 0, big[-1] if big else 0))
    print("   gap time by the kernel AFTER the gap:", [(k, round(v)) for k, v in after.most_common(6)])
    print("   gap time by the kernel BEFORE the gap:", [(k, round(v)) for k, v in before.most_common(6)])
