#!/usr/bin/env python3
"""Timing-only run of the dense iteration with a given build of the library (HPX_LIB_PATH): per-stage ms per
iteration from the plan's events, teacher-forced bandpowers so that ablated (wrong-result) builds still factor valid
matrices; a non-positive-pivot report is ignored.   python time_stages.py <tag> [C2|C3|C5] [niter]"""
import json
import sys
import pathlib

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))
CFG = {"C2": (64, 32, 256, 12, 0.0), "C3": (1024, 32, 512, 12, 0.0), "C5": (1024, 32, 1024, 12, 0.15),
       "C2L": (1024, 32, 256, 12, 0.0), "C3H": (512, 32, 512, 12, 0.0), "C3Q": (256, 32, 512, 12, 0.0)}


def main():
    import torch
    from hydra_pspec_amd import pspec, synthetic
    tag, name = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "C3")
    niter = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    nbl, T, N, M, frac = CFG[name]
    d = synthetic.make_baselines(N, T, M, k0=0, nbl=nbl, flag_frac=frac, dense=False)
    gb = pspec.GibbsBatch(d["vis"], d["flags"], d["fgmodes"], d["ninv_diag"], d["ps_prior"], 2 * niter, seed=5,
                          solver="dense")
    ps0 = np.broadcast_to(d["ps0"], (nbl, N)).copy()
    forced = np.broadcast_to(d["ps0"], (nbl, niter, N)).copy()
    st = None
    for rep in range(2):
        gb.plan.set_profiling(rep == 1)
        try:
            gb.run(niter, ps0=ps0, ps_forced=forced, keep=())
        except FloatingPointError:
            pass
        torch.cuda.synchronize()
        gb.iter_done = 0
    st = {k: round(v / niter, 4) for k, v in gb.plan.stage_ms().items()}
    print(json.dumps({"tag": tag, "cfg": name, "stage_ms": st, "sum": round(sum(st.values()), 4)}), flush=True)


if __name__ == "__main__":
    main()
