#!/bin/bash
# Produce the round's measurement artefacts on the GPU box (copied to profiles/ afterwards):
#   bench lines for C3 (with CPU baseline), C5, C2, N4, dpss, oqe and a 2-rank rehearsal; rocprofv3 kernel
#   stats of the C3 / C5-auto / dpss / oqe benches; PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy, L2 hits)
#   of the C3 bench, one counter set per pass; the FETCH_SIZE calibration probe.
# usage: tools/gpu_profile_round.sh <tag> [A|B|C|ABC]  e.g. r06 A (bench lines, rehearsals), r06 B (rocprofv3 kernel
# traces), r06 C (PMC passes -> profiles/pmc_traffic.json for the sources as they are, then the C3 line): each call fits
# gpurun's 20-minute limit.  C goes LAST: the traffic figure is hash-guarded against the kernel sources.
TAG=${1:-r04}
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/prof; mkdir -p $O
line() { grep -o '{"metric.*' $1 > $2; }
PART=${2:-ABC}
if [[ $PART == *A* ]]; then
# (the C3 line itself is taken LAST, after the PMC passes have re-stamped profiles/pmc_traffic.json for these sources)
timeout -k 10 300 python3 bench.py --steps 5 --warmup 1 --config C5 --no-cpu-baseline > $O/bench_c5.log 2>&1 && line $O/bench_c5.log $O/${TAG}_bench_c5.json; echo "C5 done"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 2 --config C2 --no-cpu-baseline > $O/bench_c2.log 2>&1 && line $O/bench_c2.log $O/${TAG}_bench_c2.json; echo "C2 done"
timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --config N4 --no-cpu-baseline > $O/bench_n4.log 2>&1 && line $O/bench_n4.log $O/${TAG}_bench_n4.json; echo "N4 done"
timeout -k 10 300 python3 bench.py --config dpss --steps 20 --warmup 2 > $O/bench_dpss.log 2>&1 && line $O/bench_dpss.log $O/${TAG}_bench_dpss.json; echo "dpss done"
timeout -k 10 300 python3 bench.py --config oqe --steps 5 --warmup 1 > $O/bench_oqe.log 2>&1 && line $O/bench_oqe.log $O/${TAG}_bench_oqe.json; echo "oqe done"
timeout -k 10 300 python3 bench.py --config fgmodes --steps 10 --warmup 2 > $O/bench_fgmodes.log 2>&1 && line $O/bench_fgmodes.log $O/${TAG}_bench_fgmodes.json; echo "fgmodes done"
timeout -k 10 300 python3 bench.py --config fgmodes --order 512 --steps 2 --warmup 1 > $O/bench_fgmodes512.log 2>&1 && line $O/bench_fgmodes512.log $O/${TAG}_bench_fgmodes_order512.json; echo "fgmodes order 512 done"
timeout -k 10 400 python3 bench.py --config C3 --noise dense --flag-frac 0.15 --steps 5 --warmup 1 > $O/bench_dense_flagged.log 2>&1 && line $O/bench_dense_flagged.log $O/${TAG}_bench_c3_dense_noise_flagged.json; echo "dense+flags done"
timeout -k 10 400 python3 bench.py --config C3 --noise dense --flag-frac 0.0 --steps 5 --warmup 1 > $O/bench_dense.log 2>&1 && line $O/bench_dense.log $O/${TAG}_bench_c3_dense_noise.json; echo "dense done"
timeout -k 10 400 python3 bench.py --config C3 --noise pertime-dense --flag-frac 0.10 --steps 5 --warmup 1 > $O/bench_ptd.log 2>&1 && line $O/bench_ptd.log $O/${TAG}_bench_pertime_dense_noise.json; echo "per-time dense done"
# (the 8-rank dry run of the launcher is a CPU test -- tests/test_bench_spawn.py -- and is not repeated here: its eight
# ranks import torch, and the box allows six processes with the GPU open)
HPX_BENCH_DEVICE=0 HPX_BENCH_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline --no-full-length > $O/bench_2rank.log 2>&1 && line $O/bench_2rank.log $O/${TAG}_bench_2ranks_one_gpu.json; echo "2-rank rehearsal done"
# the launcher with more ranks sharing the one GPU (the pool's process guard allows six processes with the GPU open,
# the launcher's own children are counted while they start: four; the 8-rank case is the driver's to run on a whole
# node): 4 ranks x 128 baselines
sleep 5
HPX_BENCH_DEVICE=0 HPX_BENCH_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 4 --nbl 128 --steps 10 --warmup 2 --no-cpu-baseline --no-full-length > $O/bench_4rank.log 2>&1 && line $O/bench_4rank.log $O/${TAG}_bench_4ranks_one_gpu.json; echo "4-rank rehearsal done"
fi
if [[ $PART == *B* ]]; then
kt() {   # kernel-trace stats: kt <name> <bench args...>
  local name=$1; shift
  cd /tmp
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$name -- python3 $R/bench.py "$@" > $O/kt_$name.log 2>&1
  cd $R
  cp $(ls $O/kt_$name/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats_$name.csv 2>/dev/null
  grep -o '{"metric.*' $O/kt_$name.log > $O/${TAG}_bench_${name}_under_rocprof.json
  echo "kernel stats $name done"
}
kt c3 --steps 10 --warmup 2 --no-cpu-baseline --no-full-length
kt c5_auto --config C5 --solver auto --steps 10 --warmup 2 --no-cpu-baseline --no-full-length
kt fgmodes --config fgmodes --steps 5 --warmup 1
kt fgmodes_order512 --config fgmodes --order 512 --steps 1 --warmup 1
kt c2 --config C2 --steps 20 --warmup 2 --no-cpu-baseline --no-full-length
kt dense_flagged --config C3 --noise dense --flag-frac 0.15 --steps 5 --warmup 1
kt dpss --config dpss --steps 10 --warmup 2
kt oqe --config oqe --steps 3 --warmup 1
fi
if [[ $PART != *C* ]]; then exit 0; fi
rm -rf $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 $O/calib
i=0
for set in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  cd /tmp
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/pmc$i -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-full-length > $O/pmc$i.log 2>&1
  cd $R
  echo "pmc pass $i done"
done
[ -x $R/tools/fetch_calib_probe ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $R/tools/fetch_calib_probe $R/tools/fetch_calib.hip
cd /tmp
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib -- $R/tools/fetch_calib_probe > $O/calib.log 2>&1; echo "calib rc=$?"
cd $R
python3 - "$TAG" <<'PY'
import csv, collections, glob, json, sys
tag = sys.argv[1]
O = "gpurun_out/prof"
lines = ["rocprofv3 --pmc <set> --output-format csv -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline   (C3: 1024 baselines x (32,512,12)); separate passes per counter set.",
         "Per-dispatch averages.  FETCH_SIZE / WRITE_SIZE in KB.  gfx950: FETCH_SIZE reads 1/2 of the bytes fetched -- calibrated for k_factor's own load shape (8 B/lane, 4 x 128-B segments) by tools/fetch_calib.hip, see the calibration lines at the end.", ""]
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
        lines.append("pass%d %-34s n=%2d dur_ms=%7.3f %s" % (i, k[:34], n, sum(dur[k]) / len(dur[k]) / 1e6,
                                                           " ".join("%s=%.4g" % kv for kv in cs.items())))
        kk = k.split("<")[0]
        kk = {"k_factor_wide": "k_factor", "k_factor_split": "k_factor", "k_backsolve_reg": "k_backsolve", "k_backsolve_x": "k_backsolve"}.get(kk, kk)     # (one name per stage)
        tot.setdefault(kk, {}).update(cs)
lines += ["", "FETCH_SIZE calibration (tools/fetch_calib.hip: 1 GiB = 1048576 KB streamed once per shape):"]
for f in glob.glob(O + "/calib/*/*counter_collection.csv"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if k.startswith("k_shape"):
            lines.append("  %-22s FETCH_SIZE_KB %s -> bytes / (FETCH_SIZE*1024) = %.3f" % (k, ["%.0f" % x for x in v], 1048576.0 / (sum(v) / len(v))))
for l in open(O + "/calib.log"):
    if l.startswith("rep 1"):
        lines.append("  " + l.rstrip())
open(O + "/%s_pmc_c3.txt" % tag, "w").write("\n".join(lines) + "\n")
out = {}
for kname in ("k_factor", "k_backsolve"):
    if kname in tot and "FETCH_SIZE" in tot[kname] and "WRITE_SIZE" in tot[kname]:
        t = tot[kname]
        out[kname] = {"fetch_kb": t["FETCH_SIZE"], "write_kb": t["WRITE_SIZE"],
                      "bytes_per_launch": (2 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024, "baselines": 1024,
                      "source": "profiles/%s_pmc_c3.txt" % tag,
                      "fetch_size_factor": 2.0, "fetch_size_factor_source": "tools/fetch_calib.hip on this kernel's load shape (ratio 2.000)",
                      "mfma_busy": t.get("SQ_VALU_MFMA_BUSY_CYCLES"), "sq_busy": t.get("SQ_BUSY_CYCLES"),
                      "gui_active": t.get("GRBM_GUI_ACTIVE"), "l2_hit": t.get("TCC_HIT_sum"), "l2_miss": t.get("TCC_MISS_sum")}
# the whole dense iteration: every kernel hpx_gibbs_run launches per step (names as in the trace)
step_kernels = ("k_assemble_tail", "k_assemble_edge", "k_factor", "k_backsolve", "k_fft_resid", "k_draw")
sb = 0.0
parts = {}
for kname in step_kernels:
    t = tot.get(kname)
    if t and "FETCH_SIZE" in t and "WRITE_SIZE" in t:
        parts[kname] = (2 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024
        sb += parts[kname]
if sb > 0:
    out["step"] = {"bytes_per_step": sb, "baselines": 1024, "by_kernel": parts,
                   "note": "sum of (2 FETCH_SIZE + WRITE_SIZE) x 1024 over the kernels of one dense Gibbs iteration"}
if out:
    sys.path.insert(0, ".")
    import bench           # (the hash of the kernel sources these counters were measured on: bench.py drops the
    json.dump({"source_hash": bench.kernel_source_hash(), "C3": out, "source_hash_files": list(bench.DENSE_STEP_SOURCES)}, open(O + "/pmc_traffic.json", "w"), indent=1)   # traffic figure for any other build)
print("\n".join(l for l in lines if "k_factor" in l or "k_backsolve" in l or "k_shape" in l))
PY
cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json
timeout -k 10 500 python3 bench.py > $O/bench_c3.log 2>&1 || { tail -5 $O/bench_c3.log; exit 1; }
line $O/bench_c3.log $O/${TAG}_bench_c3.json; echo "C3 done"
head -12 $O/${TAG}_kernel_stats_c3.csv | cut -c1-160
python3 -c "
import json
for c in ('c3','c5','c2','n4','dpss','oqe','fgmodes','fgmodes_order512','c3_dense_noise_flagged','c3_dense_noise','pertime_dense_noise','2ranks_one_gpu','4ranks_one_gpu'):
    try:
        d=json.load(open('$O/${TAG}_bench_%s.json'%c)); r=d['roofline']
        print(c, 'value %.4g %s ms/step %.3f roofline %s %.4g frac %.3f n_gpus %d' % (d['value'], d['unit'], d['ms_per_step'], r['unit'], r['achieved'], r['frac'], d['n_gpus']), {k: round(v,3) for k,v in d.get('stage_ms_per_step',{}).items()}, 'cpu', d.get('cpu_baseline',{}).get('value'), 'dev', d.get('pk_max_rel_dev_vs_cpu', d.get('max_rel_dev_vs_cpu')))
    except Exception as e: print(c, 'missing', e)
"
