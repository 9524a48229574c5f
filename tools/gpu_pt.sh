#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_pertime.py tests/test_gpu_chain.py -m gpu -q -x --timeout 300 -k "pertime or general or S_initial or initial" 2>&1 | tail -25
