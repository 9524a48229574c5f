#!/usr/bin/env python3
"""Stage timings of the dense iteration at arbitrary shapes: time_shapes.py <tag> nbl,T,N,M,flagfrac[,solver] ..."""
import json, sys, pathlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))
import torch
from hydra_pspec_amd import pspec, synthetic
tag = sys.argv[1]
for spec in sys.argv[2:]:
    f = spec.split(",")
    nbl, T, N, M = (int(v) for v in f[:4]); frac = float(f[4]); solver = f[5] if len(f) > 5 else "dense"
    niter = 8
    d = synthetic.make_baselines(N, T, M, k0=0, nbl=nbl, flag_frac=frac, dense=False)
    gb = pspec.GibbsBatch(d["vis"], d["flags"], d["fgmodes"], d["ninv_diag"], d["ps_prior"], 2 * niter, seed=5, solver=solver)
    ps0 = np.broadcast_to(d["ps0"], (nbl, N)).copy()
    out = None
    for rep in range(2):
        gb.plan.set_profiling(rep == 1)
        out = gb.run(niter, ps0=ps0, keep=())
        torch.cuda.synchronize()
        gb.iter_done = 0
    st = {k: round(v / niter, 4) for k, v in gb.plan.stage_ms().items()}
    print(json.dumps({"tag": tag, "shape": spec, "solver": gb.solver, "stage_ms": st, "sum": round(sum(st.values()), 4),
                      "ps_sum": float(out["signal_ps"].sum().item())}), flush=True)
    gb.close()
