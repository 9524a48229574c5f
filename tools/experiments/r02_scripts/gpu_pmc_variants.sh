#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / L2 hit counters of k_factor and k_backsolve for library variants
# usage: tools/gpu_pmc_variants.sh <lib path or "product"> ...
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/pmcv; mkdir -p $O
for lib in "$@"; do
  tag=$(basename $lib .so)
  [ "$lib" = product ] && export HPX_LIB_PATH= || export HPX_LIB_PATH=$R/$lib
  i=0
  for set in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
    i=$((i+1))
    cd /tmp
    timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $O/${tag}_p$i -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline ${BENCH_ARGS} > $O/${tag}_p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $O/${tag}_p$i.log; }
    cd $R
  done
  python3 - "$tag" <<'PY'
import csv, collections, glob, sys
tag = sys.argv[1]
tot = collections.defaultdict(dict)
for f in sorted(glob.glob("gpurun_out/pmcv/%s_p*/*/*counter_collection.csv" % tag)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k in acc:
        for c, v in acc[k].items():
            tot[k][c] = sum(v) / len(v)
        tot[k].setdefault("dur_ms", sum(dur[k]) / len(dur[k]) / 1e6)
for k in ("k_factor", "k_backsolve"):
    if k in tot:
        t = tot[k]
        gb = (2 * t.get("FETCH_SIZE", 0) + t.get("WRITE_SIZE", 0)) * 1024 / 1e9
        hit = t.get("TCC_HIT_sum", 0) / max(1.0, t.get("TCC_HIT_sum", 0) + t.get("TCC_MISS_sum", 0))
        busy = t.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1.0, 1024 * t.get("GRBM_GUI_ACTIVE", 0) / 8)
        print("%-22s %-12s dur %.3f ms  FETCH %.4g KB  WRITE %.4g KB  traffic(2F+W) %.2f GB  L2 hit %.3f  mfma_busy %.3f" %
              (tag, k, t["dur_ms"], t.get("FETCH_SIZE", 0), t.get("WRITE_SIZE", 0), gb, hit, busy))
PY
done
