#!/usr/bin/env python3
"""One frame of a single-frame run (one frame in flight) as a time line: every launch between two successive k_bgr2plab_t launches late in the trace, with its queue, start offset,
duration and the gap since the previous launch ended on the same queue.  python tools/frame_timeline.py <results.db> [which frame from the end = 3]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table' or type='view'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(db.execute(f"select d.start, d.end, s.kernel_name, d.queue_id from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    firsts = [i for i, r in enumerate(rows) if "k_bgr2plab_t" in r[2]]
    a, b = firsts[-back - 1], firsts[-back]
    t0 = rows[a][0]
    last_end = {}
    qs = {}
    busy = 0.0
    print("frame of %d launches, %.1f us from its first launch to the next frame's first" % (b - a, (rows[b][0] - t0) / 1e3))
    end_all = max(r[1] for r in rows[a:b])
    print("first start to last end %.1f us; sum of durations %.1f us" % ((end_all - t0) / 1e3, sum(r[1] - r[0] for r in rows[a:b]) / 1e3))
    for s, e, n, q in rows[a:b]:
        qi = qs.setdefault(q, len(qs))
        gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
        gap_any = (s - max(last_end.values())) / 1e3 if last_end else float("nan")
        last_end[q] = e
        name = n.split("(")[0].replace("void ", "").replace("rdk::", "")[:40]
        print("q%d %9.1f %7.1f  gap(q) %6.1f  gap(all) %6.1f  %s" % (qi, (s - t0) / 1e3, (e - s) / 1e3, gap, gap_any, name))


if __name__ == "__main__":
    main()
