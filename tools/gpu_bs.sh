#!/bin/bash
# back substitution: order sweep vs numpy (zpotrs; batch 200 = one workgroup per system, the old kernel's regime),
# kernel tests, then stage timings at C3 / C5 / N=256
mkdir -p gpurun_out
POTRF_NB=200 timeout -k 10 200 python tools/experiments/ab/potrf_sizes.py 100 300 400 524 656 2>&1 | grep -v "tile row" | grep -v amdgpu.ids | cut -c1-60
POTRF_NB=200 POTRF_NRHS=48 timeout -k 10 200 python tools/experiments/ab/potrf_sizes.py 100 524 2>&1 | grep -v "tile row" | grep -v amdgpu.ids | cut -c1-60
POTRF_NB=200 POTRF_NRHS=7 timeout -k 10 200 python tools/experiments/ab/potrf_sizes.py 100 524 2>&1 | grep -v "tile row" | grep -v amdgpu.ids | cut -c1-60
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout 200 -k "potr" > gpurun_out/bs_kernels.log 2>&1
rc=$?; echo "kernel tests rc=$rc"; tail -4 gpurun_out/bs_kernels.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/experiments/ab/time_shapes.py prod 1024,32,512,12,0 1024,32,1024,12,0.15 1024,32,256,12,0 2>&1 | grep "^{"
