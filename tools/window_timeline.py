#!/usr/bin/env python3
"""All launches in a window late in a kernel trace, by queue: python tools/window_timeline.py <results.db> [window us = 1600] -> start offset, duration, queue, how many other kernels were running at its start"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) if len(sys.argv) > 2 else 1600.0
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table' or type='view'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(db.execute(f"select d.start, d.end, s.kernel_name, d.queue_id from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
firsts = [i for i, r in enumerate(rows) if "k_bgr2plab_t" in r[2]]
a = firsts[-6]
t0 = rows[a][0]
qs = {}
sel = [r for r in rows[a:] if (r[0] - t0) / 1e3 < win]
busy = 0.0
events = sorted([(r[0], 1) for r in sel] + [(r[1], -1) for r in sel])
lvl, last, hist = 0, events[0][0], {}
for t, d in events:
    hist[lvl] = hist.get(lvl, 0) + (t - last) / 1e3
    lvl += d; last = t
print("time with k kernels running (us):", {k: round(v, 1) for k, v in sorted(hist.items())})
for s, e, n, q in sel:
    qi = qs.setdefault(q, len(qs))
    others = sum(1 for r in sel if r[0] <= s < r[1]) - 1
    name = n.split("(")[0].replace("void ", "")
    name = name[name.find("k_"):][:28] if "k_" in name else name[:28]
    print("%8.1f %7.1f  %s%s  +%d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, "      " * qi, "q%d" % qi, others, name))
