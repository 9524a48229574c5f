import ctypes, numpy as np, sys
sys.path.insert(0, '.')
from hydra_pspec_amd import hpx
L = hpx.lib()
out = np.zeros(1)
for it in (2000, 20000, 200000, 1000000, 3000000):
    hpx.check(L.hpx_mfma_f64_peak(it, out.ctypes.data_as(ctypes.c_void_p)))
    ms = 512 * 4.0 * it * 4.0 * 2048.0 / (out[0] * 1e12) * 1e3
    print(f"iters {it}: {out[0]:.2f} TFLOP/s  ({ms:.2f} ms kernel)")
