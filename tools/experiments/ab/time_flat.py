#!/usr/bin/env python3
"""Stage timings of the flat-noise step at C3 (solver="flat") with a given build (HPX_LIB_PATH), and the chain's
P(k) against the build named second (bit for bit?):  python time_flat.py <tag> [niter]"""
import json
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))


def main():
    import torch
    from hydra_pspec_amd import pspec, synthetic
    tag = sys.argv[1]
    niter = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    nbl, T, N, M = 1024, 32, 512, 12
    d = synthetic.make_baselines(N, T, M, k0=0, nbl=nbl, dense=False)
    gb = pspec.GibbsBatch(d["vis"], d["flags"], d["fgmodes"], d["ninv_diag"], d["ps_prior"], 2 * niter, seed=5, solver="flat")
    ps0 = np.broadcast_to(d["ps0"], (nbl, N)).copy()
    out = None
    for rep in range(2):
        gb.plan.set_profiling(rep == 1)
        out = gb.run(niter, ps0=ps0, keep=())
        torch.cuda.synchronize()
        gb.iter_done = 0
    st = {k: round(v / niter, 4) for k, v in gb.plan.stage_ms().items()}
    ps = out["signal_ps"].cpu().numpy()
    print(json.dumps({"tag": tag, "stage_ms": st, "sum": round(sum(st.values()), 4),
                      "ps_checksum": float(np.sum(ps * np.arange(1, ps.size + 1).reshape(ps.shape) % 7))}), flush=True)


if __name__ == "__main__":
    main()
