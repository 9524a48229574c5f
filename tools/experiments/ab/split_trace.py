#!/usr/bin/env python3
"""Per-step timing of the split factor (a library built with -DHPX_SPLIT_TRACE, HPX_LIB_PATH): system 0's parts,
stamps 0..6 of every tile column (loop top / inverse in LDS / column stored / look-ahead done / column complete /
column staged / trailing updates done), in microseconds from the kernel's first stamp."""
import os, sys, ctypes, pathlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))
import torch
from hydra_pspec_amd import hpx

nb, n, nrhs = int(os.environ.get("POTRF_NB", "64")), int(sys.argv[1]) if len(sys.argv) > 1 else 268, 32
rng = np.random.default_rng(1)
a = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
A = a @ a.conj().T + n * np.eye(n)
A = np.broadcast_to(A, (nb, n, n)).copy()
B = rng.standard_normal((nb, n, nrhs)) + 1j * rng.standard_normal((nb, n, nrhs))
dA = torch.from_numpy(A).cuda().contiguous(); dB = torch.from_numpy(B).cuda().contiguous(); dX = torch.zeros_like(dB)
info = torch.zeros(nb, dtype=torch.int32, device="cuda")
lib = hpx.lib()
for rep in range(3):
    hpx.check(lib.hpx_zpotrs_batched(nb, n, nrhs, hpx.ptr(dA), hpx.ptr(dB), hpx.ptr(dX), hpx.ptr(info), None))
torch.cuda.synchronize()
tr = np.zeros(8 * 41 * 8, dtype=np.int64)
lib.hpx_debug_split_trace.argtypes = [ctypes.c_void_p]
hpx.check(lib.hpx_debug_split_trace(tr.ctypes.data))
tr = tr.reshape(8, 41, 8).astype(float)
t0 = tr[tr > 0].min()
nct = (n + 15) // 16
parts = int((tr[:, 1, 0] > 0).sum())
print("parts", parts, "columns", nct, "total us", (tr.max() - t0) / 100)
for j in range(-1, nct):
    for w in range(parts):
        row = tr[w, j + 1]
        us = [(v - t0) / 100 if v > 0 else float("nan") for v in row[:7]]
        print("col %2d part %d  top %7.2f | inv %6.2f  stored %6.2f  ahead %6.2f  published %6.2f  Bdone %6.2f  staged %6.2f  updated %6.2f" %
              (j, w, us[0], *[u - us[0] for u in us[1:]], (row[7] - t0) / 100 - us[0] if row[7] > 0 else float("nan")))
