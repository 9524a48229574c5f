#!/bin/bash
# split factor (several workgroups per system): order sweep vs numpy at three batch sizes, then the C2 bench
mkdir -p gpurun_out
for nb in 2 16 64; do
  echo "== nb=$nb"
  POTRF_NB=$nb timeout -k 10 150 python tools/experiments/ab/potrf_sizes.py 20 35 64 132 150 256 300 330 2>&1 | grep -v amdgpu.ids | cut -c1-200 || exit 1
done
timeout -k 10 200 python bench.py --steps 20 --warmup 2 --config C2 --no-cpu-baseline > gpurun_out/split_c2.log 2>&1 || { tail -5 gpurun_out/split_c2.log; exit 1; }
python - <<'PY'
import json,re
d=json.loads(re.search(r'\{"metric.*', open('gpurun_out/split_c2.log').read()).group(0))
print('C2 value %.4g ms/step %.3f' % (d['value'], d['ms_per_step']), {k: round(v,4) for k,v in d['stage_ms_per_step'].items()}, d.get('full_length',{}).get('value'))
PY
