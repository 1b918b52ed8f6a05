#!/usr/bin/env python3
"""Device occupancy over time from a rocprofv3 --kernel-trace results.db: how much of the busiest window no kernel / one kernel / several kernels were running, and which kernels
run alone most.  python tools/prof_timeline.py <results.db> [fraction of the trace to keep around its middle, default 0.5]"""
import collections
import re
import sqlite3
import sys


_names = {}


def demangle(n):
    if n not in _names:
        m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", n)
        _names[n] = n[m.end(): m.end() + int(m.group(1))] if m else n.replace(".kd", "")
    return _names[n]


def main():
    db = sqlite3.connect(sys.argv[1])
    keep = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table' or type='view'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
    qcol = "d.queue_id" if "queue_id" in cols else "0"
    rows = list(db.execute(f"select d.start, d.end, s.kernel_name, {qcol} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    # the window: from the launch at the first to the one at the last `keep` quantile of all launches (the run's set-up and drain carry few launches)
    n = len(rows)
    lo, hi = rows[int(n * (0.5 - keep / 2))][0], rows[min(n - 1, int(n * (0.5 + keep / 2)))][0]
    ev = []
    queues = collections.Counter()
    for s, e, n, q in rows:
        queues[q] += 1
        s, e = max(s, lo), min(e, hi)
        if e > s:
            short = demangle(n)
            ev.append((s, 1, short)); ev.append((e, -1, short))
    ev.sort()
    level = collections.Counter(); alone = collections.Counter(); running = collections.Counter()
    gaps = []; after = collections.Counter(); prev_end = None
    last, depth = lo, 0
    for t, d, n in ev:
        if t > last:
            level[min(depth, 4)] += t - last
            if depth == 0:
                gaps.append(t - last)
                if prev_end: after[prev_end + " -> " + n] += t - last
            if depth == 1:
                alone[[k for k, v in running.items() if v > 0][0]] += t - last
            last = t
        depth += d; running[n] += d
        if d < 0: prev_end = n
    level[min(depth, 4)] += hi - last
    tot = hi - lo
    print("# window %.1f ms; kernels running at once: " % (tot / 1e6) + ", ".join("%s%d: %.1f %%" % (">=" if k == 4 else "", k, 100.0 * level[k] / tot) for k in range(5)))
    print("# hardware queues seen: %s" % dict(queues))
    if gaps:
        gaps.sort()
        print("# %d intervals with no kernel running: median %.1f us, 90 %% %.1f us, longest %.1f us; share of the idle time in gaps > 20 us: %.0f %%" % (len(gaps), gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3, gaps[-1] / 1e3, 100.0 * sum(g for g in gaps if g > 20000) / sum(gaps)))
        print("# idle time by (kernel that ended -> kernel that started), % of window:")
        for k, v in after.most_common(10):
            print("  %-60s %5.1f" % (k, 100.0 * v / tot))
    print("# time with exactly one kernel running, by kernel (% of window):")
    for n, v in alone.most_common(12):
        print("  %-44s %5.1f" % (n, 100.0 * v / tot))


if __name__ == "__main__":
    main()
