import sys, numpy as np
sys.path.insert(0, '/root/repo')
from hydra_pspec_amd import pspec, synthetic
nbl, T, N, M = 3, 8, 64, 6
d = synthetic.make_baselines(N, T, M, k0=7, nbl=nbl, flag_frac=0.12, dense=False)
std = pspec.gibbs_sample_with_fg_batched(d["vis"], d["flags"], d["fgmodes"], d["ninv_diag"], d["ps_prior"], ps_initial=d["ps0"], Niter=5, seed=3, solver="dense")
flt = np.broadcast_to(d["flags"][:, None, :], (nbl, T, N)).copy()
nt = np.broadcast_to(d["ninv_diag"][:, None, :], (nbl, T, N)).copy()
pt = pspec.gibbs_sample_with_fg_batched(d["vis"], flt, d["fgmodes"], nt, d["ps_prior"], ps_initial=d["ps0"], Niter=5, seed=3)
dev = np.abs(pt["signal_ps"] / std["signal_ps"] - 1)
print("max dev per iteration:", dev.max(axis=(0, 2)))
i = np.unravel_index(dev.argmax(), dev.shape); print("worst (baseline, iter, channel):", i, "prior channel?", bool(d["ps_prior"][0, i[2]] > 0))
std = pspec.gibbs_sample_with_fg_batched(d["vis"], d["flags"], d["fgmodes"], d["ninv_diag"], d["ps_prior"], ps_initial=d["ps0"], Niter=5, seed=3, solver="dense", keep=("signal_cr", "fg_amps", "chisq"))
pt = pspec.gibbs_sample_with_fg_batched(d["vis"], flt, d["fgmodes"], nt, d["ps_prior"], ps_initial=d["ps0"], Niter=5, seed=3, keep=("signal_cr", "fg_amps", "chisq"))
def relerr(a, b): return np.abs(a - b).max() / np.abs(b).max()
print("cr", relerr(pt["signal_cr"], std["signal_cr"]), "fg", relerr(pt["fg_amps"], std["fg_amps"]), "chisq", relerr(pt["chisq"], std["chisq"]), "lnpost", np.max(np.abs(pt["ln_post"] / std["ln_post"] - 1)))
