#!/bin/bash
# Two driver ranks sharing ONE GPU (rehearsal of the multi-rank output path on hardware): each rank samples its block of
# baselines and drains it; rank 0 merges the timings.  Compared with a one-rank run of the same baselines, bit for bit.
set -e
out=/tmp/drv2; rm -rf $out; mkdir -p $out
common="--synthetic 6,8,64 --Nfgmodes 4 --ps_prior_lo 0.1 --ps_prior_hi 2 --seed 5 --Niter 12 --write_Niter 4 --outputs ps --out_dir $out"
python run-hydra-pspec.py $common --dirname one > $out/one.log 2>&1
id=$(python -c "import uuid; print(uuid.uuid4().hex)")
for r in 0 1; do
  RANK=$r WORLD_SIZE=2 LOCAL_RANK=0 HYDRA_PSPEC_RUN_ID=$id python run-hydra-pspec.py $common --dirname two > $out/two_$r.log 2>&1 &
done
wait
python - <<'PY'
import json, numpy as np
out = "/tmp/drv2"
for k in range(1, 7):
    a, b = np.load(f"{out}/one/0-{k}/dps-eor.npy"), np.load(f"{out}/two/0-{k}/dps-eor.npy")
    assert a.shape == (12, 64) and np.allclose(a, b, rtol=1e-9), k      # (one rank may take the split factor: rounding)
t = json.load(open(f"{out}/two/timings.json"))
assert t["num_ranks"] == 2 and sorted(len(w["ant_pairs"]) for w in t["write_data"]) == [3, 3]
print("2-rank driver on one GPU: ok", t["rank_0_timers"])
PY
