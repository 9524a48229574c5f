#!/usr/bin/env python3
"""Time hpx_dft_batched's k_fft kernels alone (through the stage events of a short chain is not
possible with the timing-only HPX_FFT_DIAG builds, whose results are wrong): run a few transforms of
the C5 shape and let rocprofv3 --kernel-trace --stats report k_fft.  Usage:
    rocprofv3 --kernel-trace --stats -d out -- python3 tools/fft_probe.py [N] [T] [nbl]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from hydra_pspec_amd import hpx, utils

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
nbl = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
dev = torch.device("cuda", 0)
fop = torch.from_numpy(utils.fourier_operator(N)).to(dev)
x = torch.randn(nbl, T, N, dtype=torch.complex128, device=dev)
out = torch.empty_like(x)
L = hpx.lib()
for inv in (0, 1, 0, 1, 0, 1):
    hpx.check(L.hpx_dft_batched(nbl, T, N, hpx.ptr(fop), hpx.ptr(x), hpx.ptr(out), inv, hpx.stream_ptr(torch)))
torch.cuda.synchronize()
print("done")
