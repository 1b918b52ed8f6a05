#!/bin/bash
# the measurement set of a round, on the GPU box: bash tools/gpu_capture.sh <tag>   (results under gpurun_out/cap<tag>/)
tag=${1:-x}
R=$PWD
O=$R/gpurun_out/cap$tag
mkdir -p $O
cp tools/.capture_commit $O/commit.txt 2>/dev/null
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --slots 1 --no-cpu-baseline --no-configs --frames-per-step 128 > $O/bench_slots1.json 2>> $O/bench_default.err
python bench.py --slots 2 --no-cpu-baseline --no-configs --frames-per-step 128 > $O/bench_slots2.json 2>> $O/bench_default.err
SLOTS=32 python tools/size_sweep.py > $O/size_sweep.txt 2>&1
export TMPDIR=/tmp
cd /tmp
# kernel trace of the default command (fewer steps) and of one frame slot alone
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace_default -o t -- python $R/bench.py --steps 3 --warmup 1 --frames-per-step 64 --no-cpu-baseline --no-verify --no-configs > $O/trace_default.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace_slots1 -o t -- python $R/bench.py --steps 1 --warmup 1 --slots 1 --frames-per-step 8 --no-cpu-baseline --no-verify --no-configs > $O/trace_slots1.log 2>&1
# the other two video configurations of BASELINE.json (configs[2], configs[3]): kernel traces of the same command at their frame sizes
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace_720p -o t -- python $R/bench.py --frame 1280x720 --stream-seed 1 --steps 3 --warmup 1 --frames-per-step 64 --no-cpu-baseline --no-verify > $O/trace_720p.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace_4k -o t -- python $R/bench.py --frame 3840x2160 --stream-seed 4 --steps 3 --warmup 1 --frames-per-step 16 --no-cpu-baseline --no-verify > $O/trace_4k.log 2>&1
cd $R
# the reference on this box's OpenCL device against the goldens (tools/ref_on_opencl.py), and the GPU tests (their parity report)
for sct in stills poly stream repeat timing; do timeout 420 python tools/ref_on_opencl.py $sct > $O/ref_ocl_$sct.log 2>&1; done
cp gpurun_out/ref_opencl.json $O/ref_opencl.json 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest_gpu.log
cp gpurun_out/parity_report.json $O/parity_report.json 2>/dev/null
bash tools/gpu_pmc.sh cap$tag > $O/pmc.log 2>&1
for d in sq rd wr calrd calwr rd_720p wr_720p rd_4k wr_4k; do mkdir -p $O/pmc_$d; cp $(find gpurun_out/pmccap${tag}_$d -name "*.db" | head -1) $O/pmc_$d/results.db; done
ls -la $O
