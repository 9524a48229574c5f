#!/bin/bash
mkdir -p gpurun_out
for cr in 0 1; do
echo "== cross $cr"
HPX_EIGH_CROSS=$cr timeout -k 10 600 python tools/experiments/ab/eigh_check.py 1024 512 2>&1 | grep -v amdgpu.ids | grep "b0\|batch" | cut -c1-170
done
