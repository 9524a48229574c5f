#!/bin/bash
# usage (on the GPU box): bash tools/experiments/ab/run_ab.sh "S S2 C2 C3 C5"   -> gpurun_out/ab.log
set -e
mkdir -p gpurun_out
BASE=$PWD/tools/experiments/ab/libhpx_base.so
: > gpurun_out/ab.log
for c in $1; do
  HPX_LIB_PATH=$BASE timeout -k 10 200 python tools/experiments/ab/ab_factor.py base $c >> gpurun_out/ab.log 2>&1
  timeout -k 10 200 python tools/experiments/ab/ab_factor.py new $c >> gpurun_out/ab.log 2>&1
  python tools/experiments/ab/ab_factor.py --compare base new $c >> gpurun_out/ab.log 2>&1
done
