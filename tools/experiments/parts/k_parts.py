"""The C3 batch cut into P sub-batches on P streams (one host thread and one GibbsBatch each): does one part's
back substitution / transform / draw run under another part's factor?  (P = 2: VERDICT r2 item 2's experiment.)
usage: python tools/k_parts.py [P ...]"""
import sys, threading, time
import numpy as np
sys.path.insert(0, ".")
import torch
from hydra_pspec_amd import pspec, synthetic
nbl, T, N, M, K, W = 1024, 32, 512, 12, 30, 3
d = synthetic.make_baselines(N, T, M, nbl=nbl, dense=False)
ps0 = np.broadcast_to(d["ps0"], (nbl, N)).copy()


def make(lo, hi, stream):
    with torch.cuda.stream(stream):
        gb = pspec.GibbsBatch(d["vis"][lo:hi], d["flags"][lo:hi], d["fgmodes"], d["ninv_diag"][lo:hi], d["ps_prior"],
                              W + K, seed=d["seed"], solver="dense")
        gb.run(W, ps0=ps0[lo:hi])
    return gb


def timed(parts, delay=0.0):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = []
    for q, (gb, stream) in enumerate(parts):
        def work(gb=gb, stream=stream, q=q):
            if delay:
                time.sleep(q * delay)          # a phase offset between the parts' iterations
            with torch.cuda.stream(stream):
                gb.run(K)
        th.append(threading.Thread(target=work))
        th[-1].start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


for P in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]:
    cuts = [round(i * nbl / P) for i in range(P + 1)]
    parts = []
    for i in range(P):
        s = torch.cuda.Stream()
        parts.append((make(cuts[i], cuts[i + 1], s), s))
    timed(parts)
    for delay in (0.0, 0.0007, 0.0014) if P > 1 else (0.0,):
        for gb, _ in parts:
            gb.iter_done = W
        dt = timed(parts, delay) - (P - 1) * delay
        print(f"{P} stream(s), sizes {[cuts[i + 1] - cuts[i] for i in range(P)]}, start offset {delay * 1e3:.1f} ms: "
              f"{dt / K * 1e3:.3f} ms/step = {nbl * K / dt:.4g} baseline*iter/s", flush=True)
    for gb, _ in parts:
        gb.close()
