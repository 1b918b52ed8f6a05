#!/usr/bin/env python3
"""Which kernels wait for memory inside a divergent region?  A load under a condition compiles to  s_and_saveexec / global_load / s_waitcnt vmcnt(0) / s_or exec :
the wait is for ALL loads in flight, so a thread's independent loads travel one after the other (the region merge made 12 trips per launch that way).
python tools/isa_serial_loads.py <file.s ...>  (hipcc --save-temps=obj) -> per kernel: such regions, and all global loads"""
import re
import subprocess
import sys

for path in sys.argv[1:]:
    name, rows = None, {}
    lines = open(path).read().split("\n")
    depth_start = []
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name = m.group(1)
            rows[name] = [0, 0, []]
            depth_start = []
            continue
        if name is None:
            continue
        t = l.strip()
        if t.startswith("global_load") or t.startswith("flat_load") or t.startswith("buffer_load"):
            rows[name][1] += 1
        if t.startswith("s_and_saveexec") or t.startswith("s_andn2_saveexec"):
            depth_start.append(i)
        if t.startswith("s_or_b64 exec") and depth_start:
            s = depth_start.pop()
            body = [x.strip() for x in lines[s:i]]
            has_load = any(x.startswith(("global_load", "flat_load", "buffer_load")) for x in body)
            waits = any(x.startswith("s_waitcnt vmcnt(0)") or (x.startswith("s_waitcnt") and "vmcnt(0)" in x) for x in body)
            if has_load and waits and i - s < 40:
                rows[name][0] += 1
                rows[name][2].append(s + 1)
        if t.startswith("s_endpgm"):
            name = None
    for k, (n, loads, where) in sorted(rows.items(), key=lambda kv: -kv[1][0]):
        if n:
            d = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
            print("%3d serial regions / %3d loads  %s   (lines %s)" % (n, loads, d[:110], where[:8]))
