#!/bin/bash
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/prof; mkdir -p $O
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_fgmodes512 -- python3 $R/bench.py --config fgmodes --order 512 --steps 1 --warmup 1 > $O/kt_fgmodes512.log 2>&1
cd $R
cp $(ls $O/kt_fgmodes512/*/*kernel_stats.csv | head -1) $O/r04_kernel_stats_fgmodes_order512.csv
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof/r04_kernel_stats_fgmodes_order512.csv')))
for r in rows[:12]:
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(6), '%10.3f ms avg %10.1f ms total' % (float(r['AverageNs'])/1e6, float(r['TotalDurationNs'])/1e6), '%6.2f%%' % float(r['Percentage']))
PY
grep -o '"ms_per_step": [0-9.]*' $O/kt_fgmodes512.log
