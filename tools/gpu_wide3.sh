#!/bin/bash
# wide factor: order sweep vs numpy, kernel + chain parity with the wide build, then stage timings of the variants
mkdir -p gpurun_out
export HPX_LIB_PATH=$PWD/tools/experiments/ab/libhpx_w16.so
timeout -k 10 200 python tools/experiments/ab/potrf_sizes.py 132 524 652 780 1036 2>&1 | grep -v "tile row" | grep -v amdgpu.ids
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 200 -k "potr or reference_chain or short_chain" > gpurun_out/wide_kernels.log 2>&1
rc=$?; echo "kernel tests rc=$rc"; tail -4 gpurun_out/wide_kernels.log
[ $rc -eq 0 ] || exit $rc
unset HPX_LIB_PATH
bash tools/experiments/ab/run_time_variants.sh "$1" "$2"
