#!/bin/bash
for v in 0 3; do echo "variant $v (1: no inverse, 2: no outputs)"; ./tools/experiments/elim/elim16w_probe_v$v 256 200 2>&1 | grep -v amdgpu.ids; done
POTRF_NB=2 timeout -k 10 200 python tools/experiments/ab/potrf_sizes.py 20 132 268 330 2>&1 | grep -v amdgpu.ids | grep -v "tile row" | cut -c1-100
bash tools/experiments/ab/run_time_variants.sh "prod" "C2"
