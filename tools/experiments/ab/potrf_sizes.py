#!/usr/bin/env python3
"""zpotrf / zpotrs of a given library build (HPX_LIB_PATH) at a list of orders against numpy."""
import os, sys, pathlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))
import torch
from hydra_pspec_amd import hpx

def hpd(rng, nb, n, cond=1e3):
    a = rng.standard_normal((nb, n, n)) + 1j * rng.standard_normal((nb, n, n))
    q, _ = np.linalg.qr(a)
    ev = np.logspace(0, np.log10(cond), n)
    A = (q * ev[None, None, :]) @ np.conj(np.swapaxes(q, 1, 2))
    return 0.5 * (A + np.conj(np.swapaxes(A, 1, 2)))

rng = np.random.default_rng(3)
for n in [int(v) for v in sys.argv[1:]] or [132, 260, 524, 652, 780, 908, 1036]:
    nb, nrhs = int(os.environ.get("POTRF_NB", "2")), int(os.environ.get("POTRF_NRHS", "32"))
    A = hpd(rng, nb, n)
    B = rng.standard_normal((nb, n, nrhs)) + 1j * rng.standard_normal((nb, n, nrhs))
    dA = torch.from_numpy(np.ascontiguousarray(A)).cuda().contiguous(); dB = torch.from_numpy(np.ascontiguousarray(B)).cuda().contiguous()
    dL = torch.zeros_like(dA); dX = torch.zeros_like(dB)
    info = torch.zeros(nb, dtype=torch.int32, device="cuda")
    hpx.check(hpx.lib().hpx_zpotrf_batched(nb, n, hpx.ptr(dA), hpx.ptr(dL), hpx.ptr(info), None))
    L = dL.cpu().numpy(); ref = np.linalg.cholesky(A)
    eL = np.max(np.abs(L - ref)) / np.max(np.abs(ref))
    # first bad tile column
    bad = np.abs(L[0] - ref[0]) > 1e-8 * np.max(np.abs(ref))
    cols = np.where(bad.any(axis=0))[0]; rows = np.where(bad.any(axis=1))[0]
    info2 = torch.zeros(nb, dtype=torch.int32, device="cuda")
    try:
        hpx.check(hpx.lib().hpx_zpotrs_batched(nb, n, nrhs, hpx.ptr(dA), hpx.ptr(dB), hpx.ptr(dX), hpx.ptr(info2), None))
        X = dX.cpu().numpy(); eX = np.max(np.abs(X - np.linalg.solve(A, B))) / np.max(np.abs(X))
    except Exception as e:
        eX = str(e)
    if len(cols):
        nt = (n + 15) // 16
        tb = np.zeros((nt, nt), int)
        for i in range(nt):
            for j in range(i + 1):
                blk = bad[16 * i:16 * i + 16, 16 * j:16 * j + 16]
                tb[i, j] = 1 + np.isnan(L[0][16 * i:16 * i + 16, 16 * j:16 * j + 16]).any() if blk.any() else 0
        for i in range(nt):
            if tb[i].any():
                print("  tile row %2d: " % i + "".join(".xN"[v] for v in tb[i, :i + 1]))
    print(n, "L err %.2e" % eL, "X err", eX, "info", info.cpu().numpy().tolist()[:4], "first bad col", (cols[0] if len(cols) else None),
          "first bad row", (rows[0] if len(rows) else None), "X err", eX, flush=True)
