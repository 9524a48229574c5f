#!/usr/bin/env python3
"""Step trace of k_backsolve_x (a library built with -DHPX_BX_TRACE, HPX_LIB_PATH): a few dense iterations at C3, then
where the time of the last launch went for workgroups 0 .. 3: per pass the initial values, phase A (per group of four
chunks), the drain, phase B per 16-row step (finish + barrier | the waves' update).

  bash tools/experiments/ab/build_file_variant.sh bxtrace hpx_backsolve_lds -DHPX_BX_TRACE
  HPX_LIB_PATH=tools/experiments/ab/libhpx_bxtrace.so python tools/experiments/trace/bx_trace.py C3
"""
import ctypes as C
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))
CFG = {"C3": (1024, 32, 512, 12, 0.0), "C5": (1024, 32, 1024, 12, 0.15)}
REC, NW = 1024, 8


def main():
    import torch
    from hydra_pspec_amd import hpx, pspec, synthetic
    name = sys.argv[1] if len(sys.argv) > 1 else "C3"
    nbl, T, N, M, frac = CFG[name]
    niter = 3
    d = synthetic.make_baselines(N, T, M, k0=0, nbl=nbl, flag_frac=frac, dense=False)
    gb = pspec.GibbsBatch(d["vis"], d["flags"], d["fgmodes"], d["ninv_diag"], d["ps_prior"], 2 * niter, seed=5,
                          solver="dense")
    ps0 = np.broadcast_to(d["ps0"], (nbl, N)).copy()
    forced = np.broadcast_to(d["ps0"], (nbl, niter, N)).copy()
    gb.run(niter, ps0=ps0, ps_forced=forced, keep=())
    torch.cuda.synchronize()
    lib = hpx.lib()
    lib.hpx_debug_bx_trace.argtypes = [C.c_void_p]
    buf = np.zeros(4 * NW * REC, dtype=np.uint64)
    assert lib.hpx_debug_bx_trace(buf.ctypes.data) == 0
    buf = buf.reshape(4, NW, REC)
    ghz = 2.4                                    # s_memtime ticks at the shader clock
    for blk in (0, 1):
        print(f"================ workgroup {blk}")
        for wave in (0, 3, 7):
            r = buf[blk, wave]
            ids = (r >> np.uint64(32)).astype(np.int64)
            t = (r & np.uint64(0xffffffff)).astype(np.int64)
            n = int(np.nonzero(ids == 0)[0][0]) if (ids == 0).any() else REC
            ids, t = ids[:n], ((t[:n] - t[0]) & 0xffffffff) / ghz / 1e3
            print(f"--- wave {wave}: {n} records, total {t[-1]:.1f} us")
            kind, arg = ids >> 8, ids & 255
            i = 0
            while i < n:
                if kind[i] == 1:
                    J = arg[i]
                    seg = {"init": 0.0, "A": 0.0, "drain": 0.0, "Bfinish": 0.0, "Bupdate": 0.0, "end": 0.0}
                    groups, steps = [], []
                    j = i + 1
                    last = t[i]
                    while j < n and kind[j] != 1:
                        dt = t[j] - last
                        k = kind[j]
                        if k == 2: seg["init"] += dt
                        elif k == 3: seg["A"] += dt
                        elif k == 4: seg["A"] += dt; groups.append(dt)
                        elif k == 5: seg["drain"] += dt
                        elif k == 6: seg["Bupdate"] += dt
                        elif k == 7: seg["Bfinish"] += dt; steps.append(dt)
                        elif k == 8: seg["Bupdate"] += dt
                        elif k == 9: seg["end"] += dt
                        last = t[j]
                        j += 1
                    print(f"  pass J={J}: " + "  ".join(f"{k} {v:6.1f}" for k, v in seg.items()) +
                          (f"   groups of 4 chunks: {np.mean(groups):.2f} us x {len(groups)}" if groups else "") +
                          (f"   step (to barrier exit) {np.mean(steps):.2f} us x {len(steps)}" if steps else ""))
                    i = j
                else:
                    i += 1


if __name__ == "__main__":
    main()
