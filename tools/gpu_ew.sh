#!/bin/bash
# wide factor with the one-wave elimination (HPX_ELIM_WAVE) against the shipped build
mkdir -p gpurun_out
bash tools/experiments/ab/run_time_variants.sh "prod elimwave prod elimwave" "C3" || exit 1
bash tools/experiments/ab/run_time_variants.sh "prod elimwave" "C5" || exit 1
