#!/bin/bash
# the whole GPU suite, then short benches of C3 and C5
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 400 > gpurun_out/full_tests.log 2>&1
rc=$?; echo "gpu tests rc=$rc"; tail -5 gpurun_out/full_tests.log
[ $rc -eq 0 ] || exit $rc
for c in C3 C5; do
timeout -k 10 300 python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-full-length > gpurun_out/full_bench_$c.log 2>&1
grep -o '{"metric.*' gpurun_out/full_bench_$c.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c value %.0f  ms/step %.3f  frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac'])); print({k: round(v,3) for k,v in d['stage_ms_per_step'].items()})" || tail -5 gpurun_out/full_bench_$c.log
done
