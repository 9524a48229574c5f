#!/bin/bash
# back substitution: order sweep vs numpy (zpotrs), kernel tests, then stage timings at C3 / C2 / C5
mkdir -p gpurun_out
timeout -k 10 200 python tools/experiments/ab/potrf_sizes.py 8 17 100 132 260 272 400 524 528 640 656 1036 2>&1 | grep -v "tile row" | grep -v amdgpu.ids
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout 200 -k "potr" > gpurun_out/bs_kernels.log 2>&1
rc=$?; echo "kernel tests rc=$rc"; tail -4 gpurun_out/bs_kernels.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/experiments/ab/time_shapes.py prod 1024,32,512,12,0 64,32,256,12,0 1024,32,256,12,0 1024,203,120,12,0 2>&1 | grep "^{"
