#!/bin/bash
# bench (C3, short) for a list of library variants: tools/gpu_variants.sh product kc14 kc22 ...
for v in "$@"; do
  [ "$v" = product ] && export HPX_LIB_PATH= || export HPX_LIB_PATH=$PWD/hydra_pspec_amd/variants/libhpx_$v.so
  for rep in 1 2; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '{"metric.*' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print('%-10s factor %.3f backsolve %.3f step %.3f' % ('$v', s['factor'], s['backsolve'], d['ms_per_step']))"
  done
done
