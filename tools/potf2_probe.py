import sys, numpy as np, torch
sys.path.insert(0, '.')
from hydra_pspec_amd import hpx
rng = np.random.default_rng(0)
for n, nb in ((32, 256), (32, 512), (64, 512), (128, 512)):
    a = rng.standard_normal((nb, n, n)) + 1j * rng.standard_normal((nb, n, n))
    A = a @ np.conj(np.swapaxes(a, 1, 2)) + n * np.eye(n)
    dA = torch.from_numpy(A).cuda(); dL = torch.zeros_like(dA); info = torch.zeros(nb, dtype=torch.int32, device="cuda")
    for _ in range(3):
        hpx.check(hpx.lib().hpx_zpotrf_batched(nb, n, hpx.ptr(dA), hpx.ptr(dL), hpx.ptr(info), None))
    print("done", n, nb)
