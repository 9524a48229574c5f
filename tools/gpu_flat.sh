#!/bin/bash
# flat-noise path with the solve fused into k_fft_resid: parity tests, then the bench's flat leg fused / unfused
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x --timeout 300 -k "flat or auto or config_1 or reference_chain or statistical" > gpurun_out/flat_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/flat_tests.log
[ $rc -eq 0 ] || exit $rc
for mode in fused unfused; do
  if [ $mode = unfused ]; then export HPX_FLAT_UNFUSED=1; fi
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-full-length > gpurun_out/flat_$mode.log 2>&1 || { tail -5 gpurun_out/flat_$mode.log; exit 1; }
  python - $mode <<'PY'
import json,re,sys
d=json.loads(re.search(r'\{"metric.*', open(f'gpurun_out/flat_{sys.argv[1]}.log').read()).group(0))
f=d['flat_noise_structured_solve']
print(sys.argv[1], 'flat value %.4g ms/step %.4f dev vs dense %.2e' % (f['value'], f['ms_per_step'], f['pk_max_rel_dev_vs_dense']), {k: round(v,4) for k,v in f['stage_ms_per_step'].items()}, 'frac %.3f' % f['roofline']['frac'])
PY
done
