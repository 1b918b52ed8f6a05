#!/usr/bin/env python3
"""Durations of the successive launches of one kernel on one hardware queue (e.g. the rounds of the region merge within a group):
python tools/prof_sequence.py <results.db> <kernel name part> [how many = 40]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2]
    many = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table' or type='view'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(db.execute(f"select d.start, d.end, s.kernel_name, d.queue_id from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    rows = rows[len(rows) // 2:]
    q = rows[0][3]
    out, prev_other = [], True
    for s, e, n, qq in rows:
        if qq != q:
            continue
        if pat in n:
            out.append(("| " if prev_other and out else "") + "%.0f" % ((e - s) / 1e3))
            prev_other = False
        else:
            prev_other = True
        if len(out) >= many:
            break
    print(" ".join(out))


if __name__ == "__main__":
    main()
