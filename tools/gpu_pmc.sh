#!/bin/bash
# hardware-counter passes (own runs, kernel trace only): usage on the box: bash tools/gpu_pmc.sh <tag>
#  - SQ counters: one frame in flight, no graphs (per-kernel analysis)
#  - FETCH_SIZE / WRITE_SIZE (separate passes): the BENCHMARKED configuration - default slots, graphs on, adaptive round budget
#    (--no-verify: only the timed frames and their warm-up are in the capture), plus the calibration copy in the same capture
tag=${1:-x}
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp
CMD1="python $R/bench.py --steps 1 --warmup 1 --slots 1 --frames-per-step 4 --no-cpu-baseline --no-verify --no-configs"
CMD="python $R/bench.py --steps 2 --warmup 1 --frames-per-step 16 --no-cpu-baseline --no-verify --no-configs"
RD_NO_GRAPH=1 timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/pmc${tag}_sq -o sq -- $CMD1 > $R/gpurun_out/pmc${tag}_sq.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc${tag}_rd -o rd -- $CMD > $R/gpurun_out/pmc${tag}_rd.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc${tag}_wr -o wr -- $CMD > $R/gpurun_out/pmc${tag}_wr.log 2>&1
for cfg in "720p 1280x720 1 16" "4k 3840x2160 4 8"; do
  set -- $cfg
  C2="python $R/bench.py --frame $2 --stream-seed $3 --steps 2 --warmup 1 --frames-per-step $4 --no-cpu-baseline --no-verify"
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc${tag}_rd_$1 -o rd -- $C2 > $R/gpurun_out/pmc${tag}_rd_$1.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc${tag}_wr_$1 -o wr -- $C2 > $R/gpurun_out/pmc${tag}_wr_$1.log 2>&1
done
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc${tag}_calrd -o calrd -- python $R/tools/pmc_calibrate.py > $R/gpurun_out/pmc${tag}_calrd.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc${tag}_calwr -o calwr -- python $R/tools/pmc_calibrate.py > $R/gpurun_out/pmc${tag}_calwr.log 2>&1
cd $R; find gpurun_out/pmc${tag}_* -name "*.db" | head; tail -3 gpurun_out/pmc${tag}_sq.log
