"""Minimal check of the wide factor kernel through hpx_zpotrf_batched (stored matrix): tools/wide_repro.py [n ...]"""
import sys, time
import numpy as np
import torch as T
sys.path.insert(0, ".")
from hydra_pspec_amd import hpx
rng = np.random.default_rng(1)
for n in [int(x) for x in (sys.argv[1:] or ["260", "300", "524"])]:
    nb = 3
    a = rng.standard_normal((nb, n, n)) + 1j * rng.standard_normal((nb, n, n))
    q, _ = np.linalg.qr(a)
    A = (q * np.logspace(0, 3, n)[None, None, :]) @ np.conj(np.swapaxes(q, 1, 2))
    A = 0.5 * (A + np.conj(np.swapaxes(A, 1, 2)))
    dA = T.as_tensor(np.ascontiguousarray(A), device="cuda").contiguous()
    dL = T.zeros_like(dA)
    info = T.zeros(nb, dtype=T.int32, device="cuda")
    print("n", n, "launch", flush=True)
    t0 = time.time()
    hpx.check(hpx.lib().hpx_zpotrf_batched(nb, n, hpx.ptr(dA), hpx.ptr(dL), hpx.ptr(info), None))
    T.cuda.synchronize()
    L = dL.cpu().numpy()
    ref = np.linalg.cholesky(A)
    print("n", n, "done in %.2f s" % (time.time() - t0), "info", info.cpu().numpy(), "relerr", np.abs(L - ref).max() / np.abs(ref).max(), flush=True)
