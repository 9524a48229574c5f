#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout 300 -k "split_factor or sqrtm" 2>&1 | tail -12
