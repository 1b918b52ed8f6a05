#!/bin/bash
# on the GPU box: the reference launch by launch on the real OpenCL device against the reference on the serial stand-in (tools/ref_launches_on_opencl.py), under the goldens'
# arithmetic contract with the three loose builtins pinned (what tools/gpu_probe_ocl5.sh established for the planes up to the merge masks).  bash tools/gpu_probe_ocl6.sh
mkdir -p gpurun_out
export TMPDIR=/tmp
OPT="-Wf,-ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt -Wf,-include$PWD/oracle/refshim/rdcl_pins.h"
AMD_OCL_BUILD_OPTIONS_APPEND="$OPT" timeout 1500 python tools/ref_launches_on_opencl.py pinned 2>&1 | grep -v "^W\|^E" | cut -c1-2500
