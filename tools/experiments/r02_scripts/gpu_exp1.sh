#!/bin/bash
# factor time vs batch size for library variants (1 WG/CU, 2 WG/CU one round, two rounds)
for lib in "" hydra_pspec_amd/variants/libhpx_nofuse.so hydra_pspec_amd/variants/libhpx_base.so; do
  for nbl in 256 512 1024; do
    HPX_LIB_PATH=$lib timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --nbl $nbl 2>/dev/null | grep -o '{"metric.*' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print('lib=%-45s nbl=%4d factor %.3f backsolve %.3f step %.3f' % ('$lib' or 'product', $nbl, s['factor'], s['backsolve'], d['ms_per_step']))"
  done
done
