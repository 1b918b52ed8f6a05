#!/bin/bash
# on the GPU box: A/B points only (file of gpu_ab.sh lines), interleaved N times: bash tools/gpu_abonly.sh <file> [N=2]
for i in $(seq ${2:-2}); do bash tools/gpu_ab.sh < $1; done
