#!/usr/bin/env python3
"""Are two builds of the library bit-identical on the dense path?  Runs the same short chains with each
(HPX_LIB_PATH, one subprocess per build) and compares every output with array_equal.
  python tools/experiments/ab/bitcmp.py <libA.so> <libB.so> [C3|C5|C2] [nbl] [niter]"""
import os
import pathlib
import subprocess
import sys
import tempfile

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[3]
CFG = {"C2": (256, 0.0), "C3": (512, 0.0), "C5": (1024, 0.15)}


def child(out, name, nbl, niter):
    sys.path.insert(0, str(ROOT))
    from hydra_pspec_amd import pspec, synthetic
    N, frac = CFG[name]
    d = synthetic.make_baselines(N, 32, 12, k0=0, nbl=nbl, flag_frac=frac, dense=False)
    r = pspec.gibbs_sample_with_fg_batched(d["vis"], d["flags"], d["fgmodes"], d["ninv_diag"], d["ps_prior"],
                                           ps_initial=d["ps0"], Niter=niter, seed=d["seed"], solver="dense",
                                           keep=("signal_cr", "fg_amps", "chisq"), thin=niter)
    np.savez(out, **r)


def main():
    if sys.argv[1] == "--child":
        return child(sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]))
    a, b = sys.argv[1:3]
    name = sys.argv[3] if len(sys.argv) > 3 else "C3"
    nbl = sys.argv[4] if len(sys.argv) > 4 else "600"
    niter = sys.argv[5] if len(sys.argv) > 5 else "3"
    with tempfile.TemporaryDirectory() as td:
        outs = []
        for i, lib in enumerate((a, b)):
            out = f"{td}/o{i}.npz"
            subprocess.run([sys.executable, __file__, "--child", out, name, nbl, niter], check=True,
                           env=dict(os.environ, HPX_LIB_PATH=str(pathlib.Path(lib).resolve())))
            outs.append(dict(np.load(out)))
        same = True
        for k in outs[0]:
            eq = np.array_equal(outs[0][k], outs[1][k])
            same &= eq
            dev = float(np.max(np.abs(outs[0][k] - outs[1][k]))) if not eq else 0.0
            print(f"{name} nbl {nbl} x {niter} it  {k:10s} {'bit-identical' if eq else 'DIFFERENT (max abs %.3e)' % dev}")
        return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
