#!/usr/bin/env python3
"""Step trace of k_factor_wide (a library built with -DHPX_WIDE_TRACE, HPX_LIB_PATH): runs a few teacher-forced dense
iterations at C3 (or C5) and prints, for a pair of co-resident workgroups, where the time of the last launch went:
per super-block S / F / P totals, F per tile column by sub-phase, P per step kind (k-chunk steps, tail steps by CJ)
split into wait-for-chunk, barrier and compute, each next to its MFMA floor.

  bash tools/experiments/ab/build_wide_variant.sh wtrace -DHPX_WIDE_TRACE
  HPX_LIB_PATH=tools/experiments/ab/libhpx_wtrace.so python tools/experiments/trace/wide_trace.py C3 > gpurun_out/wtrace.txt
"""
import ctypes as C
import json
import pathlib
import sys
from collections import defaultdict

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))
CFG = {"C3": (1024, 32, 512, 12, 0.0), "C5": (1024, 32, 1024, 12, 0.15)}
REC = 1024


def decode(buf, blk, wave):
    r = buf[(blk * 4 + wave) * REC:(blk * 4 + wave + 1) * REC]
    ids = (r >> np.uint64(32)).astype(np.int64)
    t = (r & np.uint64(0xffffffff)).astype(np.int64)
    end = np.nonzero(ids == 0xffffffff)[0]
    n = int(end[0]) if len(end) else REC
    return ids[:n + 1], t[:n + 1]


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
    gb.plan.set_profiling(True)
    gb.run(niter, ps0=ps0, ps_forced=forced, keep=())
    torch.cuda.synchronize()
    st = {k: round(v / niter, 4) for k, v in gb.plan.stage_ms().items()}
    print("stage_ms per iteration (traced build):", json.dumps(st))
    lib = hpx.lib()
    lib.hpx_debug_wide_trace.restype = C.c_int
    lib.hpx_debug_wide_trace.argtypes = [C.c_void_p, C.c_int]
    buf = np.zeros(nbl * 4 * REC, dtype=np.uint64)
    assert lib.hpx_debug_wide_trace(buf.ctypes.data, nbl) == 0
    if len(sys.argv) > 2:
        np.savez_compressed(sys.argv[2], buf=buf[:64 * 4 * REC], buf256=buf[256 * 4 * REC:(256 + 8) * 4 * REC])

    # ---- who ran where and when
    info = []
    for b in range(nbl):
        ids, t = decode(buf, b, 0)
        hw, xcc = int(ids[1]), int(ids[2]) & 15
        cu, sh, se = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
        rt0, rt1 = int(ids[0]), int(ids[-2])
        info.append((b, xcc, se, sh, cu, rt0, rt1, int(t[0]), int(t[-2])))
    where = defaultdict(list)
    for x in info:
        where[x[1:5]].append(x)
    t_first = min(x[5] for x in info)
    dur = np.array([(x[6] - x[5]) & 0xffffffff for x in info]) * 0.01
    cyc = np.array([(x[8] - x[7]) & 0xffffffff for x in info])
    print(f"workgroups {nbl}; distinct (xcc,se,sh,cu) {len(where)}; per-workgroup wall time us: min {dur.min():.0f} "
          f"median {np.median(dur):.0f} max {dur.max():.0f}; shader clock {np.median(cyc / dur) / 1e3:.3f} GHz")
    starts = np.array([(x[5] - t_first) & 0xffffffff for x in info]) * 0.01
    print("start times us (sorted, every 64th):", np.sort(starts)[::64].round(0).tolist())
    ghz = float(np.median(cyc / dur)) / 1e3

    # a pair that started together on one CU
    pair = None
    for k, v in where.items():
        v = sorted(v, key=lambda x: x[5])
        if len(v) >= 2 and abs(((v[0][5] - v[1][5]) & 0xffffffff)) < 200:
            pair = (v[0][0], v[1][0])
            break
    print("traced pair (same CU, first round):", pair, "location", [info[pair[0]][1:5], info[pair[1]][1:5]])

    def us(c):
        return c / ghz / 1e3

    for blk in pair:
        print(f"\n================ workgroup {blk} ================")
        for wave in (0, 3):
            ids, t = decode(buf, blk, wave)
            t = (t - t[0]) & 0xffffffff
            ph, sb, a, bb = (ids >> 24) & 255, (ids >> 16) & 255, (ids >> 8) & 255, ids & 255
            n = len(ids)
            print(f"--- wave {wave}: {n} records, total {us(t[-2]):.1f} us")
            # S / F / P per super-block
            for J in range(8):
                s0 = [i for i in range(3, n) if ph[i] == 1 and sb[i] == J and bb[i] == 0]
                if not s0:
                    break
                s1 = [i for i in range(3, n) if ph[i] == 1 and sb[i] == J and bb[i] == 1][0]
                f1 = [i for i in range(3, n) if ph[i] == 2 and sb[i] == J and a[i] == 8][0]
                p1 = [i for i in range(3, n) if ph[i] == 6 and sb[i] == J and bb[i] == 1]
                p1 = p1[0] if p1 else f1
                print(f"  super-block {J}: S {us(t[s1] - t[s0[0]]):7.1f}  F {us(t[f1] - t[s1]):7.1f}  P {us(t[p1] - t[f1]):7.1f} us")
                # F by column
                rows = []
                for i in range(8):
                    g = {int(bb[k]): int(t[k]) for k in range(3, n) if ph[k] == 2 and sb[k] == J and a[k] == i}
                    if len(g) == 5:
                        rows.append([us(g[1] - g[0]), us(g[2] - g[1]), us(g[3] - g[2]), us(g[4] - g[3])])
                if rows:
                    r = np.array(rows)
                    print("     F columns 0..7 [elim | X tiles | barrier | T + trailing] us:")
                    for i, x in enumerate(r):
                        print(f"       {i}: " + " ".join(f"{v:6.2f}" for v in x))
                    print("     sum: " + " ".join(f"{v:6.1f}" for v in r.sum(0)))
                # P by step
                recs = [(int(a[k]), int(bb[k]), int(ph[k]), int(t[k])) for k in range(3, n) if ph[k] in (3, 4, 5) and sb[k] == J]
                if recs:
                    nk = 8 * J
                    steps = defaultdict(dict)
                    for g_, s_, p_, t_ in recs:
                        steps[(g_, s_)][p_] = t_
                    keys = sorted(steps)
                    agg = defaultdict(lambda: [0.0, 0.0, 0.0, 0])
                    for idx, key in enumerate(keys):
                        v = steps[key]
                        nxt = steps[keys[idx + 1]][3] if idx + 1 < len(keys) else None
                        kind = "k" if key[1] < nk else f"t{key[1] - nk}"
                        A = agg[kind]
                        A[0] += us(v[4] - v[3])
                        A[1] += us(v[5] - v[4])
                        if nxt is not None:
                            A[2] += us(nxt - v[5])
                            A[3] += 1
                    ngroups = len({k[0] for k in keys})
                    # from the last tail step's barrier to the next group's first wait: the group prologue
                    pro = [us(steps[keys[i + 1]][3] - steps[keys[i]][5]) for i in range(len(keys) - 1)
                           if keys[i + 1][0] != keys[i][0]]
                    print("     group prologues (last tail step's compute + next group's initial values) us: " +
                          " ".join(f"{v:.1f}" for v in pro))
                    print(f"     P: {ngroups} strip groups, {nk} k-steps + 8 tail steps each; per step kind: count, mean us [wait chunk | barrier | compute], MFMA floor of compute")
                    for kind in ["k"] + [f"t{c}" for c in range(8)]:
                        if kind not in agg:
                            continue
                        A = agg[kind]
                        cnt = max(A[3], 1)
                        nm = 96 if kind == "k" else 12 + (7 - int(kind[1:])) * 12
                        print(f"       {kind:>3}: {A[3]:4d}  {A[0] / cnt:6.2f} {A[1] / cnt:6.2f} {A[2] / cnt:6.2f}   floor {us(nm * 64):5.2f}   total {A[0] + A[1] + A[2]:7.1f}")


if __name__ == "__main__":
    main()
