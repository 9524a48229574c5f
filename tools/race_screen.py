#!/usr/bin/env python3
"""Race screen for the LDS-DMA staged kernel of the explicit-border low-rank solver (k_lr_schur:
global_load_lds + counted waits + raw barriers): the same batch repeatedly, outputs must be bit-identical
from run to run and agree with the dense path.  tools/race_screen.py [repeats]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from hydra_pspec_amd import pspec, synthetic
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for nbl, T, N, M, frac in ((8, 16, 96, 20, 0.05), (64, 203, 120, 12, 0.1), (128, 32, 1000, 12, 0.15), (256, 32, 1024, 12, 0.15),
                           (16, 40, 200, 6, 0.3)):
    d = synthetic.make_baselines(N, T, M, k0=7, nbl=nbl, flag_frac=frac, dense=False)
    kw = dict(ps_initial=d["ps0"], Niter=3, seed=d["seed"])
    args = (d["vis"], d["flags"], d["fgmodes"], d["ninv_diag"], d["ps_prior"])
    ref = pspec.gibbs_sample_with_fg_batched(*args, solver="dense", **kw)["signal_ps"]
    first = None
    for r in range(reps):
        out = pspec.gibbs_sample_with_fg_batched(*args, solver="lowrank-direct", **kw)["signal_ps"]
        if first is None:
            first = out
            live = ref > 1e-9 * np.median(ref)
            dev = np.max(np.abs(out[live] / ref[live] - 1))
        assert np.array_equal(out, first), f"run {r} differs from run 0 at shape {(nbl, T, N, M, frac)}"
    print(f"{(nbl, T, N, M, frac)}: {reps} runs bit-identical; max rel dev vs dense {dev:.2e}")
    assert dev < 1e-6
print("ok")
