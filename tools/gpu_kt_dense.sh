#!/bin/bash
# kernel-trace stats of the dense-noise + flags bench (Woodbury path)
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/prof; mkdir -p $O
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_dense_flagged -- python3 $R/bench.py --config C3 --noise dense --flag-frac 0.15 --steps 5 --warmup 1 > $O/kt_dense_flagged.log 2>&1
cd $R
cp $(ls $O/kt_dense_flagged/*/*kernel_stats.csv | head -1) $O/r04_kernel_stats_dense_flagged.csv
head -24 $O/r04_kernel_stats_dense_flagged.csv | cut -c1-150
