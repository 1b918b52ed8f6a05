#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats results.db as a per-kernel table (microseconds)."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    nframes = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    tot = sum(r[2] for r in rows)
    print(f"# total kernel time {tot:.1f} us over {nframes:g} frames = {tot/nframes:.1f} us/frame; {sum(r[1] for r in rows)/nframes:.0f} launches/frame")
    print(f"{'kernel':60s} {'calls/frame':>11s} {'avg us':>9s} {'us/frame':>10s} {'%':>6s}")
    for name, calls, total, avg, pct in rows:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")
        short = short[: short.index("(")] if "(" in short else short
        print(f"{short:60s} {calls/nframes:11.1f} {avg:9.2f} {total/nframes:10.1f} {pct:6.2f}")


if __name__ == "__main__":
    main()
