#!/bin/bash
# usage: tools/kres.sh <csrc file> <kernel name pattern> [extra hipcc flags]  -> the compiler's resource report
# (VGPRs, SGPRs, scratch, occupancy, LDS) for the matching kernels
f=$1; pat=$2; shift 2
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I$(dirname $f) "$@" -c $f -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A12 "Function Name: .*$pat" | grep -E "Function Name|VGPRs:|AGPRs|SGPRs:|ScratchSize|Occupancy|LDS Size|Spill"
