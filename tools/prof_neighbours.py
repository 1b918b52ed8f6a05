#!/usr/bin/env python3
"""What runs before and after a kernel on its hardware queue (to find where a runtime-made launch such as __amd_rocclr_fillBufferAligned comes from):
python tools/prof_neighbours.py <results.db> <kernel name part>"""
import collections
import re
import sqlite3
import sys


def short(n):
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", n)
    return n[m.end(): m.end() + int(m.group(1))] if m else n.replace(".kd", "")


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2]
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table' or type='view'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
    extra = ", d.grid_size_x, d.workgroup_size_x" if "grid_size_x" in cols else (", d.grid_x, d.workgroup_x" if "grid_x" in cols else ", 0, 0")
    rows = list(db.execute(f"select d.start, d.end, s.kernel_name, d.queue_id{extra} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    byq = collections.defaultdict(list)
    for r in rows: byq[r[3]].append(r)
    ctx = collections.Counter()
    for q, rs in byq.items():
        for i, r in enumerate(rs):
            if pat in r[2]:
                ctx[(short(rs[i - 1][2]) if i else "-", short(rs[i + 1][2]) if i + 1 < len(rs) else "-", r[4], r[5])] += 1
    for (a, b, g, w), n in ctx.most_common(15):
        print("%6d x  after %-24s before %-24s grid %s block %s" % (n, a, b, g, w))


if __name__ == "__main__":
    main()
