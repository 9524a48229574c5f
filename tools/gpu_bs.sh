#!/bin/bash
# back substitution: order sweep vs numpy (zpotrs), kernel tests, then stage timings at C3 / N=256 / order 400
mkdir -p gpurun_out
POTRF_NB=200 timeout -k 10 200 python tools/experiments/ab/potrf_sizes.py 100 280 300 400 524 2>&1 | grep -v "tile row" | grep -v amdgpu.ids | cut -c1-60
POTRF_NB=200 POTRF_NRHS=7 timeout -k 10 200 python tools/experiments/ab/potrf_sizes.py 100 430 524 2>&1 | grep -v "tile row" | grep -v amdgpu.ids | cut -c1-60
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout 200 -k "potr" > gpurun_out/bs_kernels.log 2>&1
rc=$?; echo "kernel tests rc=$rc"; tail -3 gpurun_out/bs_kernels.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/experiments/ab/time_shapes.py prod 1024,32,512,12,0 1024,32,256,12,0 1024,32,384,12,0 2>&1 | grep "^{" | cut -c1-230
