#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python tools/experiments/ab/eigh_check.py 1024 512 2>&1 | grep -v amdgpu.ids | grep "b0\|batch" | cut -c1-170
timeout -k 10 600 python -m pytest tests/test_gpu_fgmodes.py -m gpu -q -x --timeout 300 2>&1 | tail -2
