#!/bin/bash
# on the GPU box: the reference on the OpenCL device again, with build options appended through the AMD runtime's environment variable (the reference passes none):
# contraction off and correctly rounded divide / sqrt - the arithmetic contract of the goldens.  bash tools/gpu_probe_ocl2.sh
mkdir -p gpurun_out
for opt in "-Wf,-ffp-contract=off" "-Wf,-ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt"; do
  tag=$(echo "$opt" | tr -c 'a-z0-9' '_')
  rm -f gpurun_out/ref_opencl.json
  for s in stills poly stream; do
    AMD_OCL_BUILD_OPTIONS_APPEND="$opt" timeout 300 python tools/ref_on_opencl.py $s > gpurun_out/ref_ocl2_${tag}_$s.log 2>&1; echo "[$opt] section $s rc $?"
    grep -E "^still|^poly|^stream" gpurun_out/ref_ocl2_${tag}_$s.log | cut -c1-330
  done
  cp gpurun_out/ref_opencl.json gpurun_out/ref_opencl_$tag.json 2>/dev/null
done
# operator by operator (tools/ref_ops_on_opencl.py): the reference's own options, then the goldens' arithmetic contract
timeout 300 python tools/ref_ops_on_opencl.py default 2>&1 | grep -v "^W\|^E" | cut -c1-900
AMD_OCL_BUILD_OPTIONS_APPEND="-Wf,-ffp-contract=off" timeout 300 python tools/ref_ops_on_opencl.py contract_off 2>&1 | grep -v "^W\|^E" | cut -c1-900
AMD_OCL_BUILD_OPTIONS_APPEND="-Wf,-ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt" timeout 300 python tools/ref_ops_on_opencl.py contract_off_rounded 2>&1 | grep -v "^W\|^E" | cut -c1-900
