#!/bin/bash
# on the GPU box: the reference's planes stage by stage on the real OpenCL device against the oracle's (tools/ref_stages_on_opencl.py): with the reference's own empty options, and under
# the goldens' arithmetic contract with the three loose builtins pinned.  bash tools/gpu_probe_ocl5.sh
mkdir -p gpurun_out
timeout 600 python tools/ref_stages_on_opencl.py default 2>&1 | grep -v "^W\|^E" | cut -c1-1500
OPT="-Wf,-ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt -Wf,-include$PWD/oracle/refshim/rdcl_pins.h"
AMD_OCL_BUILD_OPTIONS_APPEND="$OPT" timeout 600 python tools/ref_stages_on_opencl.py pinned 2>&1 | grep -v "^W\|^E" | cut -c1-1500
