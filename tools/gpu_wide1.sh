#!/bin/bash
# first GPU contact of the wide factor: kernel-level parity, then the chain tests, then a short bench
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout 200 -k "potr" > gpurun_out/wide_kernels.log 2>&1
rc=$?; echo "kernel tests rc=$rc"; tail -15 gpurun_out/wide_kernels.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -m pytest tests -m gpu -q -x --timeout 300 -k "chain or fullsize or step or assemble" > gpurun_out/wide_chain.log 2>&1
rc=$?; echo "chain tests rc=$rc"; tail -8 gpurun_out/wide_chain.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/wide_bench.log 2>&1
grep -o '{"metric.*' gpurun_out/wide_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.0f  ms/step %.3f  frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac'])); print({k: round(v,3) for k,v in d['stage_ms_per_step'].items()})" || tail -5 gpurun_out/wide_bench.log
