"""BASELINE configs at full length: C3 = 1024 baselines x (32, 512, 12), 2000 iterations (default), or
C5 = 1024 x (32, 1024, 12) with 15 % flags (`soak_c3.py <solver> C5 [niter]`).
Checks that the chains stay finite and recover the injected spectrum; prints the sustained rate."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from hydra_pspec_amd import pspec, synthetic
nbl, T, N, M, niter, frac = 1024, 32, 512, 12, 2000, 0.0
if len(sys.argv) > 2 and sys.argv[2] == "C5":
    N, frac = 1024, 0.15
if len(sys.argv) > 3:
    niter = int(sys.argv[3])
d = synthetic.make_baselines(N, T, M, nbl=nbl, flag_frac=frac, dense=False)
gb = pspec.GibbsBatch(d["vis"], d["flags"], d["fgmodes"], d["ninv_diag"], d["ps_prior"], niter, seed=d["seed"],
                      solver=sys.argv[1] if len(sys.argv) > 1 else "dense")
ps0 = np.broadcast_to(d["ps0"], (nbl, N)).copy()
torch.cuda.synchronize()
t0 = time.perf_counter()
out = gb.run(niter, ps0=ps0)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ps = out["signal_ps"]
print(f"solver {gb.solver}: {niter} iterations in {dt:.2f} s = {nbl * niter / dt:.4g} baseline*iter/s; "
      f"finite {bool(torch.isfinite(ps).all())} ln_post finite {bool(torch.isfinite(out['ln_post']).all())}")
med = ps[:, 200:].median(dim=1).values.median(dim=0).values.cpu().numpy()
ratio = med / synthetic.true_pspec(N)
k = np.arange(N)
clean = np.abs(k - N // 2) > 12
print("median over baselines of the posterior median / injected P(k), outside the wedge: "
      f"min {ratio[clean].min():.3f} max {ratio[clean].max():.3f} mean {ratio[clean].mean():.3f}")
print("ln_post last/first (baseline 0):", float(out["ln_post"][0, -1]), float(out["ln_post"][0, 0]))
gb.close()
