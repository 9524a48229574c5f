#!/usr/bin/env python3
"""hpx_zheev_psd_batched against numpy.linalg.eigh: covariance-like matrices (a few dominant modes over a noise floor)
and generic positive definite ones; then the time for a large batch."""
import sys, time, ctypes, pathlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))
import torch
from hydra_pspec_amd import hpx

def run(A):
    nb, n0, _ = A.shape
    n = hpx.lib().hpx_zheev_psd_order(n0)
    dA = torch.from_numpy(np.ascontiguousarray(A)).cuda()
    w = torch.empty((nb, n), dtype=torch.float64, device="cuda")
    v = torch.empty((nb, n0, n), dtype=torch.complex128, device="cuda")
    sw = ctypes.c_int(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hpx.check(hpx.lib().hpx_zheev_psd_batched(nb, n0, hpx.ptr(dA), hpx.ptr(w), hpx.ptr(v), ctypes.byref(sw), None))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return w.cpu().numpy(), v.cpu().numpy(), sw.value, dt

rng = np.random.default_rng(5)
for n0, nb, kind in ((130, 3, "cov"), (300, 2, "cov"), (512, 2, "cov"), (512, 2, "hpd"), (1024, 2, "cov"), (520, 2, "gram")):
    if kind == "cov":        # np.cov of T > n0 samples: 12 strong modes over white noise
        T = n0 + 40
        modes = rng.standard_normal((n0, 12)) + 1j * rng.standard_normal((n0, 12))
        amp = (rng.standard_normal((nb, T, 12)) + 1j * rng.standard_normal((nb, T, 12))) * np.logspace(3, 0.5, 12)
        x = amp @ modes.T + (rng.standard_normal((nb, T, n0)) + 1j * rng.standard_normal((nb, T, n0)))
        A = np.stack([np.cov(x[b].T) for b in range(nb)])
    elif kind == "gram":     # rank-deficient: Gram matrix of a centred cube (exact null vector)
        N = n0 + 60
        x = rng.standard_normal((nb, n0, N)) + 1j * rng.standard_normal((nb, n0, N))
        x = x - x.mean(axis=1, keepdims=True)
        A = np.stack([x[b].conj() @ x[b].T / (n0 - 1) for b in range(nb)])
    else:
        q = rng.standard_normal((nb, n0, n0)) + 1j * rng.standard_normal((nb, n0, n0))
        A = q @ np.conj(np.swapaxes(q, 1, 2)) / n0 + np.eye(n0)
    A = 0.5 * (A + np.conj(np.swapaxes(A, 1, 2)))
    w, v, sw, dt = run(A)
    lam, U = np.linalg.eigh(A)
    for b in range(nb):
        order = np.argsort(-w[b])[:n0]
        wb, vb = w[b][order], v[b][:, order]
        lr = lam[b][::-1]
        ev_err = np.max(np.abs(wb - lr)) / lr[0]
        resid = np.linalg.norm(A[b] @ vb - vb * wb[None, :]) / np.linalg.norm(A[b])
        orth = np.abs(vb.conj().T @ vb - np.eye(n0)).max()
        lead = np.abs(np.sum(np.conj(U[b][:, ::-1][:, :12]) * vb[:, :12], axis=0))
        print(f"n {n0} {kind} b{b}: sweeps {sw} time {dt*1e3:.1f} ms  eval err/max {ev_err:.2e}  resid {resid:.2e}  orth {orth:.2e}  "
              f"1-overlap(lead 12) {np.max(1 - lead):.2e}", flush=True)
if len(sys.argv) > 1:
    nb, n0 = int(sys.argv[1]), int(sys.argv[2])
    T = n0 + 8
    x = rng.standard_normal((8, T, n0)) + 1j * rng.standard_normal((8, T, n0))
    A8 = np.stack([np.cov(x[b].T) for b in range(8)])
    A = np.ascontiguousarray(np.tile(A8, (nb // 8, 1, 1)))
    for rep in range(2):
        w, v, sw, dt = run(A)
        print(f"batch {nb} x order {n0}: {dt:.3f} s, {sw} sweeps", flush=True)
