#!/bin/bash
# per-stage times (C3, short bench, solver dense) for library variants
for v in "$@"; do
  [ "$v" = product ] && export HPX_LIB_PATH= || export HPX_LIB_PATH=$PWD/hydra_pspec_amd/variants/libhpx_$v.so
  for rep in 1 2; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '{"metric.*' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; f=d['flat_noise_structured_solve']; print('%-8s transform %.4f step %.3f | flat: solve %.4f transform %.4f step %.4f' % ('$v', s['transform'], d['ms_per_step'], f['stage_ms_per_step']['factor'], f['stage_ms_per_step']['transform'], f['ms_per_step']))"
  done
done
