#!/bin/bash
# timing-only ablation of the wide factor's tail steps (no wait / no staging): what perfect prefetch could buy
mkdir -p gpurun_out
bash tools/experiments/ab/run_time_variants.sh "prod tailnw prod tailnw" "C3" || exit 1
