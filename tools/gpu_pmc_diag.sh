#!/bin/bash
# Diagnostic counter passes on the C3 dense iteration (one counter set per pass, --pmc only): L1 / TLB behaviour,
# read latencies seen from L1 and from L2, VMEM instructions in flight, memory-pipe stalls -- for k_factor and
# k_backsolve.  usage: tools/gpu_pmc_diag.sh <tag>   -> gpurun_out/diag/<tag>_pmc_diag_c3.txt
TAG=${1:-r03}
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/diag; mkdir -p $O
i=0
for set in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_BUSY_avr" \
           "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_STREAMING_REQ_sum" \
           "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS"; do
  i=$((i+1))
  cd /tmp
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-full-length > $O/p$i.log 2>&1
  echo "pass $i rc=$?"
  cd $R
done
python3 - "$TAG" <<'PY'
import csv, collections, glob, sys
tag = sys.argv[1]
O = "gpurun_out/diag"
lines = ["rocprofv3 --pmc <set> -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-full-length (C3), one set per pass; per-dispatch averages", ""]
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
open(O + "/%s_pmc_diag_c3.txt" % tag, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
