#!/usr/bin/env python3
"""Keeps a round's parity report: copies gpurun_out/parity_report.json (written by the GPU tests, merged back by gpurun) to the tracked tests/parity_report.json,
stamped with the commit of this tree.  The tests themselves never write the tracked file."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", "parity_report.json"), os.path.join(ROOT, "tests", "parity_report.json")
rep = json.load(open(src))
rep["_measured_at"] = {"commit": subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip(),
                       "note": "measured by `pytest -m gpu` on an MI355X box (gpurun snapshot of this tree: the commit named here plus any uncommitted changes at that moment)"}
json.dump(rep, open(dst, "w"), indent=1, sort_keys=True)
print("wrote", dst, "sections:", [k for k in rep if not k.startswith("_")])
