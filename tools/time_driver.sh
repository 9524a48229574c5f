#!/bin/bash
# Time run-hydra-pspec.py end to end on one GPU at a BASELINE.json shape:
#   tools/time_driver.sh <tag> <Nbl,T,N> <Niter> [extra driver flags...]
# leaves gpurun_out/r06_driver_<tag>.{log,json} (timings.json + drain.json merged)
set -e
tag=$1; shape=$2; niter=$3; shift 3
out=/tmp/drv_$tag
rm -rf "$out"; mkdir -p "$out" gpurun_out
python run-hydra-pspec.py --synthetic "$shape" --Nfgmodes 12 --ps_prior_lo 0.1 --ps_prior_hi 2 --seed 7123689 \
  --Niter "$niter" --write_Niter 100 --out_dir "$out" --dirname run "$@" > gpurun_out/r06_driver_$tag.log 2>&1 || { tail -20 gpurun_out/r06_driver_$tag.log; exit 1; }
python - "$out" "$tag" "$@" <<'PY'
import json, sys, subprocess
out, tag = sys.argv[1], sys.argv[2]
t = json.load(open(f"{out}/run/timings.json"))
d = json.load(open(f"{out}/run/drain.json"))
du = subprocess.run(["du", "-sb", f"{out}/run"], capture_output=True, text=True).stdout.split()[0]
rec = {"tag": tag, "flags": sys.argv[3:], "rank_0_timers": t["rank_0_timers"], "drain": d, "tree_bytes": int(du),
       "total_over_process": t["rank_0_timers"]["total"] / t["rank_0_timers"]["process"]}
json.dump(rec, open(f"gpurun_out/r06_driver_{tag}.json", "w"), indent=1)
print(json.dumps(rec))
PY
rm -rf "$out"
