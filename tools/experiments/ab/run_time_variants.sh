#!/bin/bash
# usage (on the GPU box): bash tools/experiments/ab/run_time_variants.sh "prod w16 w8" "C3 C2"   -> gpurun_out/tv.log
mkdir -p gpurun_out
: > gpurun_out/tv.log
for c in $2; do
  for v in $1; do
    L=$PWD/tools/experiments/ab/libhpx_$v.so
    if [ "$v" = "prod" ]; then L=$PWD/hydra_pspec_amd/libhpx.so; fi
    HPX_LIB_PATH=$L timeout -k 10 150 python tools/experiments/ab/time_stages.py $v $c 2>&1 | grep -E '^\{|Error|error' >> gpurun_out/tv.log || echo "$v $c failed" >> gpurun_out/tv.log
  done
done
cat gpurun_out/tv.log
