#!/usr/bin/env python3
"""Run a few iterations of the C3 batch through the flat-noise solver, ignoring the result (for the
timing-only HPX_FLAT_DIAG builds); rocprofv3 --kernel-trace --stats reports k_solve_flat."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from hydra_pspec_amd import pspec, synthetic
nbl, T, N, M = 1024, 32, 512, 12
d = synthetic.make_baselines(N, T, M, nbl=nbl, dense=False)
gb = pspec.GibbsBatch(d["vis"], d["flags"], d["fgmodes"], d["ninv_diag"], d["ps_prior"], 6, seed=d["seed"], solver="flat")
try:
    gb.run(6, ps0=np.broadcast_to(d["ps0"], (nbl, N)).copy())
except Exception as e:          # ablated builds produce garbage
    print("ignored:", type(e).__name__)
print("done")
