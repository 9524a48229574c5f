#!/usr/bin/env python3
"""Stage times of one batch at an arbitrary shape: tools/shape_probe.py nbl T N M [flag_frac] [solver] [niter]
(e.g. the reference's test-data shape batched: 1024 203 120 12)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import time
import numpy as np
import torch
from hydra_pspec_amd import pspec, synthetic
nbl, T, N, M = (int(v) for v in sys.argv[1:5])
frac = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
solver = sys.argv[6] if len(sys.argv) > 6 else "auto"
niter = int(sys.argv[7]) if len(sys.argv) > 7 else 12
d = synthetic.make_baselines(N, T, M, nbl=nbl, flag_frac=frac, dense=False)
prior = d["ps_prior"] if N >= 64 else np.zeros((2, N))
gb = pspec.GibbsBatch(d["vis"], d["flags"], d["fgmodes"], d["ninv_diag"], prior, niter + 2, seed=d["seed"], solver=solver)
ps0 = np.broadcast_to(d["ps0"], (nbl, N)).copy()
gb.run(2, ps0=ps0)
torch.cuda.synchronize()
gb.plan.set_profiling(True)
t0 = time.perf_counter()
out = gb.run(niter)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{nbl} x (T {T}, N {N}, M {M}), flags {frac}, solver {gb.solver}: {dt / niter * 1e3:.3f} ms/step = "
      f"{nbl * niter / dt:.4g} baseline*iter/s; stages", {k: round(v / niter, 3) for k, v in gb.plan.stage_ms().items()},
      "finite", bool(torch.isfinite(out["signal_ps"]).all()))
gb.close()
