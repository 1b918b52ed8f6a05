#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace results.db: how many kernels run at once, what runs alone, per-queue busy share, and for each
kernel the time-weighted number of OTHER kernels running beside it.  usage: timeline.py results.db [skip_fraction]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4        # leading part of the trace left out (warm-up, graph capture)
rows = list(db.execute("select name, queue_id, start, end from kernels order by start"))
t0, t1 = min(r[2] for r in rows), max(r[3] for r in rows)
lo = t0 + (t1 - t0) * skip
rows = [r for r in rows if r[2] >= lo]
t0, t1 = min(r[2] for r in rows), max(r[3] for r in rows)
short = lambda n: (n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0])
ev = []
for i, (n, q, s, e) in enumerate(rows): ev.append((s, 1, i)); ev.append((e, -1, i))
ev.sort()
hist = collections.Counter(); alone = collections.Counter(); beside = collections.Counter(); dur = collections.Counter()
running = set(); last = t0
for t, d, i in ev:
    dt = t - last
    if dt > 0:
        k = len(running); hist[k] += dt
        for j in running:
            nm = short(rows[j][0]); beside[nm] += dt * (k - 1); dur[nm] += dt
            if k == 1: alone[nm] += dt
    last = t
    if d > 0: running.add(i)
    else: running.discard(i)
span = t1 - t0
print(f"span {span/1e3:.0f} us, {len(rows)} kernels")
print("kernels running at once -> share of time:", {k: round(v / span, 3) for k, v in sorted(hist.items())})
print("mean:", round(sum(k * v for k, v in hist.items()) / span, 2))
qbusy = collections.Counter()
for n, q, s, e in rows: qbusy[q] += e - s
print("busy share per queue:", {q: round(v / span, 3) for q, v in sorted(qbusy.items())})
print(f"{'kernel':40s} {'time %':>7s} {'alone %':>8s} {'others beside':>14s}")
for nm, v in sorted(dur.items(), key=lambda kv: -kv[1])[:28]:
    print(f"{nm:40s} {100*v/sum(dur.values()):7.2f} {100*alone[nm]/v:8.1f} {beside[nm]/v:14.2f}")
