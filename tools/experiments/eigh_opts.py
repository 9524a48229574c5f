#!/usr/bin/env python3
"""Order-512 covariance eigensolver (bench.py --config fgmodes --order 512) with HPX_OPT_EIGH_INNER_SWEEPS = 1, 2 and
the per-sweep convergence trace.   python tools/experiments/eigh_opts.py [order] [nb]"""
import sys
import time
import pathlib

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))


def main():
    import torch
    from hydra_pspec_amd import fgmodes, hpx
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    T, nm = N + 8, 12
    rng = np.random.default_rng(0)
    nsrc = min(nb, 64)
    vis = (rng.standard_normal((nsrc, T, N)) + 1j * rng.standard_normal((nsrc, T, N)))
    vis = np.tile(vis, (nb // nsrc, 1, 1))
    d_vis = torch.from_numpy(vis).cuda()
    for inner in (1, 2):
        hpx.set_option(hpx.OPT_EIGH_INNER_SWEEPS, inner)
        hpx.set_option(hpx.OPT_EIGH_TRACE, 1)
        fgmodes.cov_eig_modes(d_vis, nm, return_evals=True, as_numpy=False)
        torch.cuda.synchronize()
        hpx.set_option(hpx.OPT_EIGH_TRACE, 0)
        t0 = time.perf_counter()
        fgmodes.cov_eig_modes(d_vis, nm, return_evals=True, as_numpy=False)
        torch.cuda.synchronize()
        print(f"inner sweeps {inner}: {1e3 * (time.perf_counter() - t0):.1f} ms for {nb} baselines of order {N}", flush=True)
    hpx.set_option(hpx.OPT_EIGH_INNER_SWEEPS, 1)


if __name__ == "__main__":
    main()
