"""VERDICT r2 item 2's cheap experiment: the C3 batch as two half-batches on two streams (two host threads, one
GibbsBatch each), so that one half's back substitution / transform / draw can run under the other half's factor.
Prints the rate of the whole batch on one stream and of the two halves together."""
import sys, threading, time
import numpy as np
sys.path.insert(0, ".")
import torch
from hydra_pspec_amd import pspec, synthetic
nbl, T, N, M, K, W = 1024, 32, 512, 12, 20, 3
d = synthetic.make_baselines(N, T, M, nbl=nbl, dense=False)
ps0 = np.broadcast_to(d["ps0"], (nbl, N)).copy()


def make(lo, hi, stream):
    with torch.cuda.stream(stream):
        gb = pspec.GibbsBatch(d["vis"][lo:hi], d["flags"][lo:hi], d["fgmodes"], d["ninv_diag"][lo:hi], d["ps_prior"],
                              W + K, seed=d["seed"], solver="dense")
        gb.run(W, ps0=ps0[lo:hi])
    return gb


def timed(parts):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = []
    for gb, stream in parts:
        def work(gb=gb, stream=stream):
            with torch.cuda.stream(stream):
                gb.run(K)
        th.append(threading.Thread(target=work))
        th[-1].start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


s0, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
whole = make(0, nbl, s0)
dt = timed([(whole, s0)])
print(f"one stream, {nbl} baselines: {dt / K * 1e3:.3f} ms/step = {nbl * K / dt:.4g} baseline*iter/s")
whole.close()
a, b = make(0, nbl // 2, s1), make(nbl // 2, nbl, s2)
dt = timed([(a, s1), (b, s2)])
print(f"two streams, 2 x {nbl // 2} baselines: {dt / K * 1e3:.3f} ms/step = {nbl * K / dt:.4g} baseline*iter/s")
