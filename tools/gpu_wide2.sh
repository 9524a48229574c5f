#!/bin/bash
# wide factor, iteration loop: kernel parity with the wide build, then stage timings of the variants
mkdir -p gpurun_out
HPX_LIB_PATH=$PWD/tools/experiments/ab/libhpx_w16.so timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 200 -k "potr or reference_chain or short_chain" > gpurun_out/wide_kernels.log 2>&1
rc=$?; echo "kernel tests rc=$rc"; tail -4 gpurun_out/wide_kernels.log
[ $rc -eq 0 ] || exit $rc
bash tools/experiments/ab/run_time_variants.sh "$1" "$2"
