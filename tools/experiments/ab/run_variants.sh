#!/bin/bash
# usage (on the GPU box): bash tools/experiments/ab/run_variants.sh "base v1 v2" "C2 C3"   -> gpurun_out/abv.log
set -e
mkdir -p gpurun_out
: > gpurun_out/abv.log
for c in $2; do
  for v in $1; do
    L=$PWD/tools/experiments/ab/libhpx_$v.so
    if [ "$v" = "prod" ]; then L=$PWD/hydra_pspec_amd/libhpx.so; fi
    HPX_LIB_PATH=$L timeout -k 10 200 python tools/experiments/ab/ab_factor.py $v $c >> gpurun_out/abv.log 2>&1
    if [ "$v" != "base" ]; then python tools/experiments/ab/ab_factor.py --compare base $v $c | grep -v "True" >> gpurun_out/abv.log 2>&1 || true; fi
  done
done
rm -f gpurun_out/ab_*.npz
