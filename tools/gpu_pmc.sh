#!/bin/bash
# collect PMC passes for the bench (2 steps), print per-kernel averages for k_factor
export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out/pmc
i=0
for set in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc/p$i -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline > gpurun_out/pmc/p$i.log 2>&1
done
python3 - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob("gpurun_out/pmc/p*/*/*counter_collection.csv")):
    acc=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")[:10]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    for k in acc:
        if k.startswith("k_factor") or k.startswith("k_backsolv"):
            print(k, "dur_ms %.3f" % (sum(dur[k])/len(dur[k])/1e6), {c: "%.4g" % (sum(v)/len(v)) for c,v in acc[k].items()})
PY
