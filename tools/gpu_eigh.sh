#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_fgmodes.py -m gpu -q -x --timeout 300 2>&1 | tail -4
