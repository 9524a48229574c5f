#!/bin/bash
for n in "$@"; do
  echo "=== nbl $n"
  timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --nbl $n 2>/dev/null | grep -o '{"metric.*' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.0f  ms/step %.2f  factor TF %.1f' % (d['value'], d['ms_per_step'], d['roofline']['achieved'])); print({k: round(v,3) for k,v in d['stage_ms_per_step'].items()})"
done
