#!/bin/bash
# Produce the round's measurement artefacts on the GPU box (copied to profiles/ afterwards):
#   bench lines for C3 (with CPU baseline), C5, C2; rocprofv3 kernel stats of the C3 bench;
#   PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy) of the C3 bench, one counter set per pass.
# usage: tools/gpu_profile_round.sh <tag>      e.g. r01
TAG=${1:-r01}
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/prof; mkdir -p $O
timeout -k 10 400 python3 bench.py > $O/bench_c3.log 2>&1 || { tail -5 $O/bench_c3.log; exit 1; }
grep -o '{"metric.*' $O/bench_c3.log > $O/${TAG}_bench_c3.json; echo "C3 done"
timeout -k 10 300 python3 bench.py --steps 5 --warmup 1 --config C5 --no-cpu-baseline > $O/bench_c5.log 2>&1 && grep -o '{"metric.*' $O/bench_c5.log > $O/${TAG}_bench_c5.json; echo "C5 done"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 2 --config C2 --no-cpu-baseline > $O/bench_c2.log 2>&1 && grep -o '{"metric.*' $O/bench_c2.log > $O/${TAG}_bench_c2.json; echo "C2 done"
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/kt.log 2>&1
cd $R
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats_c3.csv 2>/dev/null; echo "kernel stats done"
# the structured (flagged, flat-noise) path at C5: per-kernel times of solver=auto
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt5 -- python3 $R/bench.py --config C5 --solver auto --steps 10 --warmup 2 --no-cpu-baseline > $O/kt5.log 2>&1
cd $R
cp $(ls $O/kt5/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats_c5_auto.csv 2>/dev/null
grep -o '{"metric.*' $O/kt5.log > $O/${TAG}_bench_c5_auto_under_rocprof.json; echo "C5 auto kernel stats done"
i=0
for set in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  cd /tmp
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/pmc$i -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline > $O/pmc$i.log 2>&1
  cd $R
  echo "pmc pass $i done"
done
python3 - "$TAG" <<'PY'
import csv, collections, glob, json, sys
tag = sys.argv[1]
O = "gpurun_out/prof"
lines = ["rocprofv3 --pmc <set> --output-format csv -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline   (C3: 1024 baselines x (32,512,12)); separate passes per counter set.",
         "Per-dispatch averages.  FETCH_SIZE / WRITE_SIZE in KB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts TCC_EA0_RDREQ x 64 B while requests are 128 B wide -> double it.", ""]
tot = {}
for i, f in enumerate(sorted(glob.glob(O + "/pmc*/*/*counter_collection.csv")), 1):
    acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k in sorted(acc):
        cs = {c: sum(v) / len(v) for c, v in acc[k].items()}
        n = len(next(iter(acc[k].values())))
        lines.append("pass%d %-24s n=%2d dur_ms=%7.3f %s" % (i, k, n, sum(dur[k]) / len(dur[k]) / 1e6,
                                                           " ".join("%s=%.4g" % kv for kv in cs.items())))
        tot.setdefault(k, {}).update(cs)
open(O + "/%s_pmc_c3.txt" % tag, "w").write("\n".join(lines) + "\n")
kf = [k for k in tot if k.startswith("k_factor")]
if kf and "FETCH_SIZE" in tot[kf[0]] and "WRITE_SIZE" in tot[kf[0]]:
    t = tot[kf[0]]
    json.dump({"C3": {"k_factor": {"fetch_kb": t["FETCH_SIZE"], "write_kb": t["WRITE_SIZE"],
                                   "bytes_per_launch": (2 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024,
                                   "baselines": 1024, "source": "profiles/%s_pmc_c3.txt" % tag,
                                   "mfma_busy": t.get("SQ_VALU_MFMA_BUSY_CYCLES"), "sq_busy": t.get("SQ_BUSY_CYCLES"),
                                   "gui_active": t.get("GRBM_GUI_ACTIVE")}}},
              open(O + "/pmc_traffic.json", "w"), indent=1)
print("\n".join(l for l in lines if "k_factor" in l or "k_backsolve" in l))
PY
head -8 $O/${TAG}_kernel_stats_c3.csv | cut -c1-150
python3 -c "
import json
for c in ('c3','c5','c2'):
    try:
        d=json.load(open('$O/${TAG}_bench_%s.json'%c)); print(c, 'value %.4g ms/step %.3f factor TF %.1f frac %.3f'%(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac']), {k: round(v,3) for k,v in d['stage_ms_per_step'].items()}, 'cpu', d.get('cpu_baseline',{}).get('value'), 'dev', d.get('pk_max_rel_dev_vs_cpu'))
    except Exception as e: print(c, 'missing', e)
"
