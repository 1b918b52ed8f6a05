#!/bin/bash
# on the GPU box: the reference on the OpenCL device under the goldens' arithmetic contract AND with the three loosely specified builtins pinned to the stand-in's definitions
# (oracle/refshim/rdcl_pins.h forced into the reference's programs): operator by operator, then end to end.  bash tools/gpu_probe_ocl3.sh
mkdir -p gpurun_out
OPT="-Wf,-ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt -Wf,-include$PWD/oracle/refshim/rdcl_pins.h"
AMD_OCL_BUILD_OPTIONS_APPEND="$OPT" timeout 300 python tools/ref_ops_on_opencl.py pinned 2>&1 | grep -v "^W\|^E" | cut -c1-900
rm -f gpurun_out/ref_opencl.json
for s in stills poly stream repeat; do
  AMD_OCL_BUILD_OPTIONS_APPEND="$OPT" timeout 300 python tools/ref_on_opencl.py $s > gpurun_out/ref_ocl3_$s.log 2>&1; echo "section $s rc $?"
  grep -E "^still|^poly|^stream|^repeat|rror" gpurun_out/ref_ocl3_$s.log | cut -c1-330
done
cp gpurun_out/ref_opencl.json gpurun_out/ref_opencl_pinned.json 2>/dev/null
