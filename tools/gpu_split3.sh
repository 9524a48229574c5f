#!/bin/bash
# split factor: order sweep (light and forced-heavy hand-off), stage timings, per-step trace
mkdir -p gpurun_out
POTRF_NB=2 timeout -k 10 200 python tools/experiments/ab/potrf_sizes.py 20 132 268 330 524 2>&1 | grep -v amdgpu.ids | grep -v "tile row" | cut -c1-160 || exit 1
POTRF_NB=30 HPX_SPLIT_HEAVY=1 timeout -k 10 200 python tools/experiments/ab/potrf_sizes.py 35 268 300 2>&1 | grep -v amdgpu.ids | grep -v "tile row" | cut -c1-100 || exit 1
POTRF_NB=64 timeout -k 10 200 python tools/experiments/ab/potrf_sizes.py 35 268 300 2>&1 | grep -v amdgpu.ids | grep -v "tile row" | cut -c1-100 || exit 1
bash tools/experiments/ab/run_time_variants.sh "prod" "C3 C2" || exit 1
HPX_LIB_PATH=$PWD/tools/experiments/ab/libhpx_trace.so timeout -k 10 150 python tools/experiments/ab/split_trace.py 268 2>&1 | grep -v amdgpu.ids > gpurun_out/split_trace.txt
head -34 gpurun_out/split_trace.txt
