export TMPDIR=/tmp RD_NO_FORK=1 RD_DIAG_NO_POST=1
R=$PWD
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/profconc -o rc -- python $R/bench.py --steps 2 --warmup 1 --slots 16 --frames-per-step 32 --no-cpu-baseline > $R/gpurun_out/profconc.log 2>&1
cd $R; tail -1 gpurun_out/profconc.log | cut -c1-100
