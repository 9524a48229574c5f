#!/bin/bash
# the whole GPU suite, then a short bench of the metric's configuration
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 400 > gpurun_out/full_tests.log 2>&1
rc=$?; echo "gpu tests rc=$rc"; tail -5 gpurun_out/full_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-full-length > gpurun_out/full_bench.log 2>&1
grep -o '{"metric.*' gpurun_out/full_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.0f  ms/step %.3f  frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac'])); print({k: round(v,3) for k,v in d['stage_ms_per_step'].items()})" || tail -5 gpurun_out/full_bench.log
