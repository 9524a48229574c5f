#!/usr/bin/env python3
"""Per-phase wall-clock trace of k_factor's block-column loop (workgroup 0) from the traced library
(make_trace_variant.py).  usage: HPX_LIB_PATH=.../libhpx_trace.so python trace_factor.py [C2|C3|C5] [nbl]"""
import ctypes as C
import sys
import pathlib

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[3]))
from hydra_pspec_amd import hpx, pspec, synthetic  # noqa: E402

CFG = {"C2": (64, 32, 256, 12, 0.0), "C3": (1024, 32, 512, 12, 0.0), "C5": (1024, 32, 1024, 12, 0.15)}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C2"
    nbl, T, N, M, frac = CFG[name]
    if len(sys.argv) > 2:
        nbl = int(sys.argv[2])
    d = synthetic.make_baselines(N, T, M, k0=0, nbl=nbl, flag_frac=frac, dense=False)
    out = pspec.gibbs_sample_with_fg_batched(d["vis"], d["flags"], d["fgmodes"], d["ninv_diag"], d["ps_prior"],
                                             ps_initial=d["ps0"], Niter=3, seed=1, solver="dense")
    assert np.isfinite(out["signal_ps"]).all()
    lib = hpx.lib()
    buf = (C.c_ulonglong * (4 * 64 * 16))()
    lib.hpx_trace_read.restype = C.c_int
    assert lib.hpx_trace_read(buf) == 0
    t = np.array(buf, dtype=np.int64).reshape(4, 64, 16)         # [wave][column][id], 10 ns ticks
    ncol = int((t[0, :, 9] > 0).sum())
    names = ["(enter)", "tail", "D->lds+sync", "elimA", "scale+B,C", "elimD", "scale+E+sync", "groups", "partial", "barrier"]
    print(f"{name}: nbl={nbl}, {ncol} block columns; microseconds per phase, wave 0 | max over waves")
    tot = np.zeros(10)
    for j in range(ncol):
        row = []
        prev = t[:, j - 1, 9] if j > 0 else t[:, j, 0]
        seq = [t[:, j, i] for i in range(10)]
        last = prev.copy()
        for i in range(10):
            cur = seq[i].copy()
            bad = cur <= 0
            cur[bad] = last[bad]
            dt = (cur - last) / 100.0
            row.append(f"{dt[0]:6.2f}|{dt.max():6.2f}")
            tot[i] += dt.max()
            last = cur
        print(f"j={j:2d} " + " ".join(row) + f"   col {(t[0, j, 9] - (t[0, j - 1, 9] if j > 0 else t[0, j, 0])) / 100.0:7.2f}")
    print("phase:", " ".join(f"{n:>13s}" for n in names))
    print("sum  :", " ".join(f"{v:13.2f}" for v in tot), f"  total {(t[0, ncol - 1, 9] - t[0, 0, 0]) / 100.0:.1f} us")


if __name__ == "__main__":
    main()
