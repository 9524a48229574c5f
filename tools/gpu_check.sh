#!/bin/bash
# quick GPU iteration: unit parity of the factor/solve kernels, then a short bench
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout 300 -k "potr or dft or assemble" > gpurun_out/chk_tests.log 2>&1
tail -2 gpurun_out/chk_tests.log
timeout -k 10 300 python bench.py --steps ${1:-5} --warmup 1 --no-cpu-baseline ${2:-} > gpurun_out/chk_bench.log 2>&1
grep -o '{"metric.*' gpurun_out/chk_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.0f  ms/step %.2f  factor TF %.1f' % (d['value'], d['ms_per_step'], d['roofline']['achieved'])); print({k: round(v,3) for k,v in d['stage_ms_per_step'].items()}); print('peak_measured', d['roofline']['peak_measured'])" || tail -5 gpurun_out/chk_bench.log
