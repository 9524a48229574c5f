#!/bin/bash
# quick GPU iteration: parity tests that exercise the factor/solve kernels and the chains, then a short C3 bench
# usage: tools/gpu_quick.sh [pytest -k expression] [bench args...]
mkdir -p gpurun_out
K=${1:-"potr or chain or fullsize or step"}; shift
timeout -k 10 600 python -m pytest tests -m gpu -q -x --timeout 300 -k "$K" > gpurun_out/quick_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/quick_tests.log
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/quick_bench.log 2>&1
grep -o '{"metric.*' gpurun_out/quick_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.0f  ms/step %.3f  roofline %s %.1f frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['unit'], d['roofline']['achieved'], d['roofline']['frac'])); print({k: round(v,3) for k,v in d['stage_ms_per_step'].items()})" || tail -5 gpurun_out/quick_bench.log
