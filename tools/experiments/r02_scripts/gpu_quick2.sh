#!/bin/bash
# parity of the factor path + bench at three batch sizes, for the environment given by the caller
timeout -k 10 500 python -m pytest tests -m gpu -q -x --timeout 400 -k "potr or chain or fullsize or step or lowrank" 2>&1 | tail -3
for n in 256 512 1024; do timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --nbl $n 2>/dev/null | grep -o '{"metric.*' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print('nbl=%4d factor %.3f backsolve %.3f step %.3f value %.0f' % ($n, s['factor'], s['backsolve'], d['ms_per_step'], d['value']))"; done
