#!/usr/bin/env python3
"""A/B of two builds of the library on the dense Gibbs iteration: run as
   HPX_LIB_PATH=<lib> python ab_factor.py <tag> [C2|C3|C5|...]  -> gpurun_out/ab_<tag>_<cfg>.npz (P(k), signal, timings)
and compare two runs with  python ab_factor.py --compare tagA tagB cfg."""
import json
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))
OUT = ROOT / "gpurun_out"
CFG = {"C2": (64, 32, 256, 12, 0.0), "C3": (1024, 32, 512, 12, 0.0), "C5": (1024, 32, 1024, 12, 0.15),
       "C3F": (1024, 32, 512, 12, 0.1), "C2L": (1024, 32, 256, 12, 0.0), "N64": (2048, 32, 64, 6, 0.0), "S": (8, 16, 112, 6, 0.1), "S2": (5, 8, 100, 4, 0.0)}


def run(tag, name):
    import torch
    from hydra_pspec_amd import pspec, synthetic
    nbl, T, N, M, frac = CFG[name]
    d = synthetic.make_baselines(N, T, M, k0=0, nbl=nbl, flag_frac=frac, dense=False)
    niter = 10
    gb = pspec.GibbsBatch(d["vis"], d["flags"], d["fgmodes"], d["ninv_diag"], d["ps_prior"], 3 * niter, seed=5,
                          solver="dense")
    out = gb.run(niter, ps0=np.broadcast_to(d["ps0"], (nbl, N)).copy(), keep=("signal_cr",), thin=niter)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out2 = gb.run(niter, keep=())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / niter
    gb.plan.set_profiling(True)
    gb.run(niter, keep=())
    torch.cuda.synchronize()
    st = {k: v / niter for k, v in gb.plan.stage_ms().items()}
    OUT.mkdir(exist_ok=True)
    cr = out["signal_cr"].cpu().numpy()
    np.savez(OUT / f"ab_{tag}_{name}.npz", ps=out["signal_ps"].cpu().numpy()[:, -1], cr=cr[::max(1, nbl // 4), 0, :2],
             crsum=np.array([np.abs(cr).sum()]), ps2=out2["signal_ps"].cpu().numpy()[:, -1])
    print(json.dumps({"tag": tag, "cfg": name, "ms_per_iter": round(dt * 1e3, 4),
                      "stage_ms": {k: round(v, 4) for k, v in st.items()}}))


def compare(a, b, name):
    A, B = np.load(OUT / f"ab_{a}_{name}.npz"), np.load(OUT / f"ab_{b}_{name}.npz")
    for k in ("ps", "cr", "crsum", "ps2"):
        same = np.array_equal(A[k], B[k])
        rel = np.max(np.abs(A[k] - B[k])) / np.max(np.abs(A[k]))
        print(f"{name} {k}: bit-identical={same} max rel diff={rel:.3e}")


if __name__ == "__main__":
    if sys.argv[1] == "--compare":
        compare(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        run(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "C3")
