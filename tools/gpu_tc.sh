#!/bin/bash
# k_fft_resid with 16 instead of 8 time columns per workgroup (one workgroup per CU): stage timings, parity of a short chain
mkdir -p gpurun_out
bash tools/experiments/ab/run_time_variants.sh "prod tc16 prod tc16" "C3" || exit 1
HPX_LIB_PATH=$PWD/tools/experiments/ab/libhpx_tc16.so timeout -k 10 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x --timeout 200 -k "short_chain" 2>&1 | tail -2
