#!/usr/bin/env python3
"""For every device-wide idle interval longer than <min us> in the middle of a rocprofv3 --kernel-trace results.db: per hardware queue, the kernel that ended last before it
and the one that starts next after it (with the distance in us).  python tools/prof_gaps.py <results.db> [min us = 40] [how many = 12]"""
import re
import sqlite3
import sys


def short(n):
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", n)
    return n[m.end(): m.end() + int(m.group(1))] if m else n.replace(".kd", "")


def main():
    db = sqlite3.connect(sys.argv[1])
    mn = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 40e3
    many = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table' or type='view'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(db.execute(f"select d.start, d.end, s.kernel_name, d.queue_id from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    n = len(rows)
    rows = rows[n // 4: 3 * n // 4]
    t0 = rows[0][0]
    end = rows[0][1]
    shown = 0
    for i in range(1, len(rows)):
        s = rows[i][0]
        if s - end > mn and shown < many:
            shown += 1
            print("idle %.0f us at t = %.2f ms" % ((s - end) / 1e3, (end - t0) / 1e6))
            for q in sorted(set(r[3] for r in rows)):
                before = [r for r in rows[:i] if r[3] == q]
                after = [r for r in rows[i:] if r[3] == q]
                b = before[-1] if before else None
                a = after[0] if after else None
                print("   queue %s: %-22s ended %7.0f us before the gap's end | next %-22s starts %7.0f us after the gap's start" % (
                    q, short(b[2]) if b else "-", (s - b[1]) / 1e3 if b else 0, short(a[2]) if a else "-", (a[0] - end) / 1e3 if a else 0))
        end = max(end, rows[i][1])


if __name__ == "__main__":
    main()
