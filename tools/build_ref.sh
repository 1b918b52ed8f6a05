#!/bin/bash
# the library of another commit as a tuning build (run HERE): bash tools/build_ref.sh <commit> <name>  -> rectdetect_amd/variants/lib<name>.so (selected by RD_LIB_PATH)
set -e
cd "$(dirname "$0")/.."
c=$1; name=$2
rm -rf /tmp/rd_ref_$name && git worktree prune && git worktree add -f /tmp/rd_ref_$name $c > /dev/null 2>&1
python /tmp/rd_ref_$name/tools/gen_luts.py > /dev/null 2>&1 || true
make -s -j8 -C /tmp/rd_ref_$name/rectdetect_amd/csrc > /dev/null
mkdir -p rectdetect_amd/variants
cp /tmp/rd_ref_$name/rectdetect_amd/librectdetect_hip.so rectdetect_amd/variants/lib$name.so
git worktree remove --force /tmp/rd_ref_$name
echo built rectdetect_amd/variants/lib$name.so from $(git rev-parse --short $c)
