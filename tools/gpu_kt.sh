#!/bin/bash
# kernel-trace stats of a bench run: tools/gpu_kt.sh <name> <bench args...>  -> gpurun_out/kt_<name>.csv (+ top lines on stdout)
name=$1; shift
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$name -- python3 $R/bench.py "$@" > $O/kt_$name.log 2>&1
cd $R
cp $(ls $O/kt_$name/*/*kernel_stats.csv | head -1) $O/kt_$name.csv 2>/dev/null
python3 - $O/kt_$name.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print("%-60s calls %5s avg_us %10.1f total_ms %9.2f pct %5s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
grep -o '{"metric.*' $O/kt_$name.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print(d.get('stage_ms_per_step'))"
