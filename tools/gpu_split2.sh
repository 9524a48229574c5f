#!/bin/bash
# split factor: per-step trace of one launch (debugging build)
export HPX_LIB_PATH=$PWD/tools/experiments/ab/libhpx_trace.so
timeout -k 10 150 python tools/experiments/ab/split_trace.py 268 2>&1 | grep -v amdgpu.ids
