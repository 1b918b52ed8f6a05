#!/bin/bash
# on the GPU box: 3840x2160 streams with and without group launches (side_config of bench.py with a chosen number of frames in flight): bash tools/gpu_4k_groups.sh
for cfg in "16 -" "32 8" "16 -" "32 8" "32 4"; do
  set -- $cfg
  if [ "$2" = "-" ]; then unset RD_ZBATCH; else export RD_ZBATCH=$2; fi
  python - $1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import bench, rectdetect_amd as ra
slots = int(sys.argv[1])
L = ra.lib()
c = bench.side_config(ra, L, "4K", 3840, 2160, 4, 32, slots, 0)
print("slots", slots, "RD_ZBATCH", os.environ.get("RD_ZBATCH", "-"), c["value"], "frames/s, per launch", c["frames_per_launch"], "repeated", c["frames_repeated"])
PY
done
