#!/bin/bash
# on the GPU box: the reference on the OpenCL device over the four long streams of the parity table (516 frames), (a) with its own empty options, (b) under the goldens' arithmetic
# contract with the three loose builtins pinned (tools/gpu_probe_ocl3.sh).  bash tools/gpu_probe_ocl4.sh
mkdir -p gpurun_out
rm -f gpurun_out/ref_opencl.json
timeout 900 python tools/ref_on_opencl.py streamlong 2>&1 | grep "^streamlong"
cp gpurun_out/ref_opencl.json gpurun_out/ref_opencl_long_default.json
rm -f gpurun_out/ref_opencl.json
OPT="-Wf,-ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt -Wf,-include$PWD/oracle/refshim/rdcl_pins.h"
AMD_OCL_BUILD_OPTIONS_APPEND="$OPT" timeout 900 python tools/ref_on_opencl.py streamlong 2>&1 | grep "^streamlong"
cp gpurun_out/ref_opencl.json gpurun_out/ref_opencl_long_pinned.json
