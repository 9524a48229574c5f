#!/bin/bash
# correlated-noise paths: parity tests, then the dense + flags bench row
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 400 -k "correlated or dense or pertime or gcr or noise" > gpurun_out/dense_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/dense_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --config C3 --noise dense --flag-frac 0.15 --steps 5 --warmup 1 > gpurun_out/bench_dense_flagged.log 2>&1 || { tail -5 gpurun_out/bench_dense_flagged.log; exit 1; }
timeout -k 10 600 python bench.py --config C3 --noise pertime-dense --flag-frac 0.10 --steps 5 --warmup 1 > gpurun_out/bench_pertime_dense.log 2>&1 || { tail -5 gpurun_out/bench_pertime_dense.log; exit 1; }
python - <<'PY'
import json,re
for n in ("dense_flagged", "pertime_dense"):
    d=json.loads(re.search(r'\{"metric.*', open(f'gpurun_out/bench_{n}.log').read()).group(0))
    print(n, 'value %.4g ms/step %.3f setup %.2f s' % (d['value'], d['ms_per_step'], d['setup_seconds']), {k: round(v,3) for k,v in d['stage_ms_per_step'].items()}, 'frac %.3f' % d['roofline']['frac'])
PY
