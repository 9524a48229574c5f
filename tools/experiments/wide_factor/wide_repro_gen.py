"""Minimal check of the wide factor kernel in closed-form (generator) mode: a short dense chain at N = 512."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from hydra_pspec_amd import pspec, synthetic
nbl, T, N, M = 2, 32, 512, 12
d = synthetic.make_baselines(N, T, M, k0=11, nbl=nbl, dense=False)
kw = dict(ps_initial=d["ps0"], Niter=2, seed=d["seed"])
print("dense launch", flush=True)
t0 = time.time()
a = pspec.gibbs_sample_with_fg_batched(d["vis"], d["flags"], d["fgmodes"], d["ninv_diag"], d["ps_prior"], solver="dense", **kw)
print("dense done %.2f s" % (time.time() - t0), flush=True)
b = pspec.gibbs_sample_with_fg_batched(d["vis"], d["flags"], d["fgmodes"], d["ninv_diag"], d["ps_prior"], solver="flat", **kw)
print("max rel dev vs flat", np.max(np.abs(b["signal_ps"] / a["signal_ps"] - 1)), flush=True)
