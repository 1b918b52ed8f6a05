#!/usr/bin/env python3
"""which hardware queues a kernel trace used: python tools/trace_queues.py <results.db>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table' or type='view'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
print(cols)
for r in db.execute(f"select queue_id, stream_id, count(*), min(start), max(end) from {kd} group by queue_id, stream_id order by 1, 2"):
    print(r)
