import sys, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo')
import torch
from hydra_pspec_amd import hpx, pspec, synthetic
nbl, T, N, M = 64, 32, 256, 12
d = synthetic.make_baselines(N, T, M, k0=0, nbl=nbl, flag_frac=0.0, dense=False)
gb = pspec.GibbsBatch(d["vis"], d["flags"], d["fgmodes"], d["ninv_diag"], d["ps_prior"], 8, seed=5, solver="dense")
ps0 = np.broadcast_to(d["ps0"], (nbl, N)).copy()
gb.run(4, ps0=ps0)
torch.cuda.synchronize()
lib = hpx.lib()
buf = np.zeros(8 * 64, dtype=np.uint64)
lib.hpx_debug_bs_trace.argtypes = [C.c_void_p]
assert lib.hpx_debug_bs_trace(buf.ctypes.data) == 0
b = buf.reshape(8, 64).astype(np.int64)
for w in (0, 3, 7):
    t = b[w]
    t0 = t[0]
    print("wave", w, "prologue wait us %.2f" % ((t[1] - t[0]) / 2400.0), "end drain %.2f" % ((t[61] - t[60]) / 2400.0), "total %.2f" % ((t[61] - t[0]) / 2400.0))
    for J in range(16, 0, -1):
        a, bb, c = t[2 + 3 * J], t[3 + 3 * J], t[4 + 3 * J]
        nxt = t[2 + 3 * (J - 1)] if J > 1 else t[60]
        print("  J=%2d barrier %.2f  wait %.2f  work %.2f" % (J, (bb - a) / 2400.0, (c - bb) / 2400.0, (nxt - c) / 2400.0))
