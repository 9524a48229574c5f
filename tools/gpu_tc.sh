#!/bin/bash
# k_fft_resid with 4 instead of 8 time columns per workgroup (four workgroups per CU): stage timings, parity of a short chain
mkdir -p gpurun_out
bash tools/experiments/ab/run_time_variants.sh "prod tc4 prod tc4" "C3" || exit 1
HPX_LIB_PATH=$PWD/tools/experiments/ab/libhpx_tc4.so timeout -k 10 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x --timeout 200 -k "short_chain" 2>&1 | tail -2
