"""GibbsParts (hpx_gibbs_run_parts) against one GibbsBatch at C3: ms per iteration for 1..5 parts."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from hydra_pspec_amd import pspec, synthetic
nbl, T, N, M, K, W = 1024, 32, 512, 12, 50, 3
d = synthetic.make_baselines(N, T, M, nbl=nbl, dense=False)
ps0 = np.broadcast_to(d["ps0"], (nbl, N)).copy()
ref = None
for P in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 5]:
    cls = dict(parts=P) if P > 1 else {}
    gb = (pspec.GibbsParts if P > 1 else pspec.GibbsBatch)(d["vis"], d["flags"], d["fgmodes"], d["ninv_diag"],
                                                             d["ps_prior"], W + 4 * K, seed=d["seed"], solver="dense", **cls)
    gb.run(W, ps0=ps0)
    torch.cuda.synchronize()
    dts = []
    for rep in range(4):
        t0 = time.perf_counter()
        out = gb.run(K)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        dts.append(round(dt / K * 1e3, 3))
        if rep == 0:
            first = out["signal_ps"]
    if ref is None:
        ref = first
    print(f"{P} part(s): {dt / K * 1e3:.3f} ms/iteration = {nbl * K / dt:.4g} baseline*iter/s; same chain: "
          f"{bool(torch.equal(first, ref))}; ms/iteration of the four runs of {K}: {dts}", flush=True)
    gb.close()
