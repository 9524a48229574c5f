#!/bin/bash
# wide factor with the one-wave elimination (HPX_ELIM_WAVE) against the shipped build
mkdir -p gpurun_out
HPX_LIB_PATH=$PWD/tools/experiments/ab/libhpx_elimwave.so POTRF_NB=600 timeout -k 10 200 python tools/experiments/ab/potrf_sizes.py 524 652 1036 2>&1 | grep -v amdgpu.ids | grep -v "tile row" | cut -c1-80 || exit 1
bash tools/experiments/ab/run_time_variants.sh "prod elimwave prod elimwave" "C3" || exit 1
bash tools/experiments/ab/run_time_variants.sh "prod elimwave" "C5" || exit 1
