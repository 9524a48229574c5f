#!/usr/bin/env python3
"""Phase times of k_fft_resid (library built with -DHPX_FR_TRACE, HPX_LIB_PATH) for workgroups 0 .. 7 of the last launch
of a few C3 iterations: load issue | loads landed + LDS writes | FFT passes | tiles | reduction.
  bash tools/experiments/ab/build_file_variant.sh frtrace hpx_post -DHPX_FR_TRACE
  HPX_LIB_PATH=tools/experiments/ab/libhpx_frtrace.so python tools/experiments/trace/fr_trace.py"""
import ctypes as C
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))


def main():
    import torch
    from hydra_pspec_amd import hpx, pspec, synthetic
    nbl, T, N, M = 1024, 32, 512, 12
    d = synthetic.make_baselines(N, T, M, k0=0, nbl=nbl, flag_frac=0.0, dense=False)
    gb = pspec.GibbsBatch(d["vis"], d["flags"], d["fgmodes"], d["ninv_diag"], d["ps_prior"], 6, seed=5, solver="dense")
    ps0 = np.broadcast_to(d["ps0"], (nbl, N)).copy()
    gb.run(3, ps0=ps0, keep=())
    torch.cuda.synchronize()
    lib = hpx.lib()
    lib.hpx_debug_fr_trace.argtypes = [C.c_void_p]
    buf = np.zeros(8 * 4 * 16, dtype=np.uint64)
    assert lib.hpx_debug_fr_trace(buf.ctypes.data) == 0
    t = buf.reshape(8, 4, 16).astype(np.int64)
    names = ["set-up -> loads issued", "loads landed, LDS written", "barrier", "pass 1", "pass 2", "pass 3 + barrier",
             "tiles", "reduction"]
    for blk in range(8):
        for w in (0, 3):
            r = t[blk, w]
            seg = [(r[1] - r[0]), (r[2] - r[1]), 0, (r[3] - r[2]), (r[4] - r[3]), (r[6] - r[4]), (r[7] - r[6]), (r[8] - r[7])]
            print(f"wg {blk} wave {w}: " + "  ".join(f"{n} {v / 2400.0:5.2f}" for n, v in zip(names, seg) if n != "barrier") +
                  f"   total {(r[8] - r[0]) / 2400.0:6.2f} us")


if __name__ == "__main__":
    main()
