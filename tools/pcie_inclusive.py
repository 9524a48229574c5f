"""PCIe-inclusive rate of the batched entry point: host numpy arrays in, host numpy arrays out
(plan set-up, RNG tables, H2D of the visibilities, the chain, D2H of P(k) and ln-posterior)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from hydra_pspec_amd import pspec, synthetic
nbl, T, N, M = 1024, 32, 512, 12
d = synthetic.make_baselines(N, T, M, nbl=nbl, dense=False)
for niter in (10, 100):
    for solver in ("dense", "auto"):
        pspec.gibbs_sample_with_fg_batched(d["vis"][:8], d["flags"][:8], d["fgmodes"], d["ninv_diag"][:8], d["ps_prior"],
                                           ps_initial=d["ps0"], Niter=2, seed=1, solver=solver)      # warm-up
        t0 = time.perf_counter()
        out = pspec.gibbs_sample_with_fg_batched(d["vis"], d["flags"], d["fgmodes"], d["ninv_diag"], d["ps_prior"],
                                                 ps_initial=d["ps0"], Niter=niter, seed=1, solver=solver)
        dt = time.perf_counter() - t0
        print(f"solver={solver:5s} Niter={niter:4d}: {dt:.3f} s wall, {nbl * niter / dt:.4g} baseline*iter/s "
              f"(host arrays in/out, {d['vis'].nbytes / 1e6:.0f} MB of visibilities up, "
              f"{out['signal_ps'].nbytes / 1e6:.0f} MB of P(k) down)")
