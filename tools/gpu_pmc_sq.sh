#!/bin/bash
# Wave-state counters of the C3 dense iteration (SQ block only, one set per pass): where the waves of k_factor_wide /
# k_backsolve spend their cycles -- parked at a wait or barrier (SQ_WAIT_ANY), stalled at issue (SQ_WAIT_INST_ANY),
# issuing (SQ_ACTIVE_INST_*).  usage: tools/gpu_pmc_sq.sh <tag> [library]   -> gpurun_out/sq/<tag>_pmc_sq_c3.txt
TAG=${1:-r05}
[ -n "$2" ] && export HPX_LIB_PATH=$2
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/sq; mkdir -p $O; rm -rf $O/p*
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
           "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  cd /tmp
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/tools/experiments/ab/time_stages.py pmc C3 3 > $O/p$i.log 2>&1
  echo "pass $i rc=$?"
  cd $R
done
python3 - "$TAG" <<'PY'
import csv, collections, glob, sys
tag = sys.argv[1]
O = "gpurun_out/sq"
lines = ["rocprofv3 --pmc <SQ set> -- python3 tools/experiments/ab/time_stages.py pmc C3 3, one set per pass; per-dispatch averages", ""]
for i, f in enumerate(sorted(glob.glob(O + "/p*/*/*counter_collection.csv")), 1):
    acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if not (k.startswith("k_factor") or k.startswith("k_backsolve") or k.startswith("k_fft_resid")):
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k in sorted(acc):
        cs = {c: sum(v) / len(v) for c, v in acc[k].items()}
        lines.append("%-22s dur_ms=%7.3f %s" % (k[:22], sum(dur[k]) / len(dur[k]) / 1e6, " ".join("%s=%.5g" % kv for kv in sorted(cs.items()))))
open(O + "/%s_pmc_sq_c3.txt" % tag, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
