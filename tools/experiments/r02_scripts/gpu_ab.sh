#!/bin/bash
# A/B two builds of libhpx on the GPU box: $1, $2 = make variable assignments (e.g. HPX_RT3=0 HPX_RT3=1)
mkdir -p gpurun_out
for v in "$@"; do
  make -C hydra_pspec_amd/csrc clean > /dev/null; make -C hydra_pspec_amd/csrc -j8 $v > gpurun_out/ab_build.log 2>&1 || { tail -5 gpurun_out/ab_build.log; exit 1; }
  echo "=== $v"
  [ -z "$NOTEST" ] && timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout 300 -k "potr" 2>&1 | tail -1
  timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | grep -o '{"metric.*' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.0f  ms/step %.2f  factor TF %.1f' % (d['value'], d['ms_per_step'], d['roofline']['achieved'])); print({k: round(v,3) for k,v in d['stage_ms_per_step'].items()})"
done
