#!/bin/bash
# on the GPU box: is there an OpenCL device, and what does the reference do on it (tools/ref_on_opencl.py)
mkdir -p gpurun_out
clinfo 2>&1 | grep -E "Number of platforms|Platform Name|Number of devices|Device Type|Board name|  Name:|Max compute units|Device OpenCL C version|Driver version" | head -20 > gpurun_out/clinfo.txt
cat gpurun_out/clinfo.txt
for s in stills poly stream repeat timing; do
  timeout 420 python tools/ref_on_opencl.py $s > gpurun_out/ref_ocl_$s.log 2>&1; echo "section $s rc $?" | tee -a gpurun_out/ref_ocl_rc.log
  tail -12 gpurun_out/ref_ocl_$s.log
done
