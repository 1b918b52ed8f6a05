#!/bin/bash
# tuning builds (run HERE): bash tools/variants.sh name "-DRR_PX=4 -DRR_TY=8" ...   -> rectdetect_amd/variants/lib<name>.so, selected at run time by RD_LIB_PATH
set -e
cd "$(dirname "$0")/../rectdetect_amd/csrc"
mkdir -p ../variants
name=$1; shift
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DCL_TARGET_OPENCL_VERSION=120 -Wno-unused-result -DRD_TUNING -I. -I../../include"
mkdir -p build/var_$name
for f in rd_k_rect rd_k_label rd_k_front rd_k_nms; do x=""; [ $f = rd_k_nms ] && x="-fno-slp-vectorize"; hipcc $FL $x "$@" -c $f.hip -o build/var_$name/$f.o & done; wait
hipcc $FL "$@" -c rd_api.hip -o build/var_$name/rd_api.o      # (-DRD_TUNING: the switches of the experiments, rd_api.hip)
objs=""; for f in rd_k_poly rd_k_post rd_runtime rd_post rd_helper rd_synth; do objs="$objs build/$f.o"; done; objs="$objs build/var_$name/rd_api.o"
hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/lib$name.so $objs build/var_$name/rd_k_rect.o build/var_$name/rd_k_label.o build/var_$name/rd_k_front.o build/var_$name/rd_k_nms.o -lpthread -lm
echo built ../variants/lib$name.so
