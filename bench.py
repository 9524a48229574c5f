#!/usr/bin/env python3
"""Benchmark of the MI355X Gibbs hot path on BASELINE.json's metric:
baseline x Gibbs-iterations per second at Nfreq = 512.

    python bench.py --gpus N --steps K --warmup W

One "step" = one Gibbs iteration of every baseline resident on the GPU
(assemble -> Cholesky factor + forward solve -> back solve -> back transform ->
residual / chi^2 / ln-posterior -> bandpower draw).  Workload (config C3 of
BASELINE.json / SURVEY 8d): 1024 synthetic baselines per GPU of shape
(Ntimes, Nfreq, Nfgmodes) = (32, 512, 12), DPSS foreground modes, 7-bin prior.
N > 1 ranks each own a contiguous block of 1024 more baselines (C4 at N = 8):
independent chains, no collective on the data path ("weak" scaling); rank 0
reports (total baselines x K) / (max over ranks of the K-step wall time).
Without a launcher (`WORLD_SIZE` unset) `--gpus N` makes this process a launcher
that starts the N ranks itself and never touches a GPU.
Inputs are resident in HBM when the timed region starts.

Extra objects on the JSON line: ``roofline`` for the dominant kernel (k_factor:
FP64-MFMA bound; duration from HIP events recorded on the launch stream inside
the timed region) and ``cpu_baseline`` (the numpy/scipy oracle = operation-by-
operation port of the reference, timed on this host on a bounded sample).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

CONFIGS = {
    # name: (baselines per GPU, Ntimes, Nfreq, Nmodes, flag fraction)
    "C2": (64, 32, 256, 12, 0.0),
    "C3": (1024, 32, 512, 12, 0.0),
    "C5": (1024, 32, 1024, 12, 0.15),
    # N4: time-dependent flags, one system per (baseline, time): 32 x 32 = 1024
    # factorisations of the C3 order per iteration
    "N4": (32, 32, 512, 12, 0.10),
}
# iterations BASELINE.json names per config (C5: "iteration count as C3", cut to 500 on the dense path to bound
# the run time -- SURVEY 8d allows it when stated); the `full_length` leg runs them in one call
FULL_ITERS = {"C2": 1000, "C3": 2000, "C5": 500}
FP64_MFMA_PEAK_TFLOPS = 78.6   # AMD spec (32 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz); the local
#                                microarch guide has no f64 row -- the measured issue peak is
#                                reported next to it as roofline.peak_measured.
HBM_PEAK_GBS = 8000.0


def flops_factor(N, M, T):
    """Algorithmic real flops of k_factor per unit (one baseline x one iteration):
    complex Cholesky of the (N+M) system + forward substitution of T right-hand sides."""
    n = N + M
    return 4.0 / 3.0 * n ** 3 + 4.0 * n * n * T


def flops_unit(N, M, T):
    """SURVEY 8(d) F_alg per unit for the whole iteration."""
    n = N + M
    return 4.0 / 3.0 * n ** 3 + 8.0 * n * n * T + 2 * T * 5 * N * np.log2(N) + 8.0 * N * N


def bytes_unit(N, M, T):
    """SURVEY 8(d) B_alg per unit."""
    n = N + M
    return 16.0 * N * N + 2 * 16.0 * n * n + 3 * 16.0 * T * n + 8.0 * N


def cpu_baseline(N, T, M, flag_frac, nbl=2, niter=3):
    """Time the oracle (port of the reference's numpy/scipy path) on this host."""
    from oracle import pspec_ref
    from hydra_pspec_amd import synthetic
    d = synthetic.make_baselines(N, T, M, k0=0, nbl=nbl, flag_frac=flag_frac)
    chains = []
    t0 = time.perf_counter()
    for b in range(nbl):
        r = pspec_ref.gibbs_sample_with_fg(d["vis"][b], d["flags"][b], d["S_initial"], d["fgmodes"],
                                           d["Ninv"], d["ps_prior"], Niter=niter, seed=d["seed"])
        chains.append(r[2])
    dt = time.perf_counter() - t0
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count()
    return dict(value=nbl * niter / dt, unit="baseline*iter/s", cores=int(cores), kind="port",
                sample=f"{nbl} baselines x {niter} iterations of (Ntimes {T}, Nfreq {N}, Nmodes {M}), "
                       f"one process, default BLAS threads, {dt:.1f} s"), np.array(chains), d


def host_cores():
    """Cores this process may use: the affinity mask, cut to the cgroup CPU quota where one is set."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    quota = None
    try:       # cgroup v2: "<quota> <period>" or "max <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(np.ceil(int(q) / int(per))))
    except Exception:
        pass
    return (min(avail, quota) if quota else avail), avail, quota


def cpu_baseline_multiproc(N, T, M, flag_frac, niter=8, max_procs=None):
    """P independent single-BLAS-thread processes, one baseline stream each (the
    reference's MPI deployment model), aggregate rate, on every core this process is allowed to use
    (affinity mask and cgroup quota; `max_procs` / HPX_BENCH_CPU_PROCS caps it).  Child processes are plain
    `python -m oracle.cpu_worker` runs: they never touch the GPU."""
    import subprocess
    usable, avail, quota = host_cores()
    if max_procs is None and os.environ.get("HPX_BENCH_CPU_PROCS"):
        max_procs = int(os.environ["HPX_BENCH_CPU_PROCS"])
    P = max(1, min(usable, max_procs) if max_procs else usable)
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1",
               HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_worker", str(N), str(T), str(M), str(flag_frac),
                               str(k), str(niter)], cwd=str(REPO), env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.DEVNULL, text=True) for k in range(P)]
    secs = []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            pr.kill()
            continue
        for line in out.splitlines():
            if line.startswith("CPU_WORKER_SECONDS"):
                secs.append(float(line.split()[1]))
    wall = time.perf_counter() - t0
    if not secs:
        return None
    return dict(value=len(secs) * niter / max(secs), unit="baseline*iter/s", processes=len(secs), blas_threads=1,
                host_cores=os.cpu_count(), affinity_cores=avail, cgroup_quota_cores=quota,
                sample=f"{len(secs)} single-thread processes (of {os.cpu_count()} host cores; affinity mask {avail}, "
                       f"cgroup quota {quota if quota else 'none'}"
                       f"{', capped at ' + str(max_procs) if max_procs else ''}) x 1 baseline x {niter} iterations, "
                       f"slowest {max(secs):.1f} s, wall incl. start-up {wall:.1f} s")


DENSE_STEP_SOURCES = ("hpx_factor.hip", "hpx_factor_wide.hip", "hpx_factor_split.hip", "hpx_factor_tiles.h",
                      "hpx_backsolve.hip", "hpx_backsolve_lds.hip", "hpx_plan.hip", "hpx_setup.hip", "hpx_chain.hip", "hpx_post.hip", "hpx_woodbury.hip", "hpx_chain.h",
                      "hpx_transform.hip", "hpx_internal.h", "hpx_fft.h", "Makefile")


def kernel_source_hash():
    """Hash of the sources the kernels of the dense Gibbs iteration are built from (k_assemble_tail, k_factor,
    k_backsolve, k_fft_resid, k_draw): profiles/pmc_traffic.json carries the hash it was measured on, and a
    library built from other sources gets no `roofline.traffic` from it."""
    import hashlib
    h = hashlib.sha256()
    for name in DENSE_STEP_SOURCES:
        f = REPO / "hydra_pspec_amd" / "csrc" / name
        h.update(name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def flops_flat(N, M, T):
    """Structured solve without flags (k_solve_flat): two MFMA contractions over the channels and the
    back product, DESIGN.md section 4."""
    return 8.0 * N * M * (M + 2 * T)


def bytes_flat(N, M, T):
    """k_solve_flat per unit: reads the invariant blocks Q (N x T) and G (N x M), writes X ((N+M) x T); c128."""
    return 16.0 * N * (T + M) + 16.0 * (N + M) * T


def bytes_fft_resid(N, M, T):
    """k_fft_resid per unit: reads the solution z and the data d (N x T c128 each), writes the per-block
    partial sums of |z|^2 (N doubles per block of 8-16 time columns)."""
    return 2 * 16.0 * N * T + 8.0 * N * max(1, T // max(1, min(16, 4096 // N)))


def flops_lowrank(N, M, T, f):
    """Low-rank solve, FFT form: transforms of the border + the (M+f) Cholesky
    and its two triangular solves."""
    n = M + f
    return 5.0 * N * np.log2(N) * (1 + M + 2 * T) + 4.0 / 3.0 * n ** 3 + 8.0 * n * n * T + 8.0 * N * M * (M + 2 * T)


def roofline_for(solver, stage, nbl, N, M, T, fmax, K, traffic, peak_meas):
    """Roofline object of the dominant kernel (stage) of the solver that actually ran.  Durations are
    the HIP-event stage times of the timed region (hpx_plan_stage_ms), averaged per launch."""
    if solver == "dense":
        ms = stage["factor"] / K
        fl = flops_factor(N, M, T)
        ach = nbl * fl / (ms * 1e-3) / 1e12
        import ctypes
        from hydra_pspec_amd import hpx
        parts = ctypes.c_int(0)
        form = hpx.lib().hpx_factor_form(nbl, N + M, T, ctypes.byref(parts))
        name = {0: "k_factor (32-wide block columns)", 1: "k_factor_wide (128-column super-blocks, LDS-staged panels)",
                2: f"k_factor_split ({parts.value} co-operating workgroups per system, tiles in registers)"}[form]
        return {"kernel": name + ": batched complex Cholesky + forward solve, FP64 MFMA", "bound": "mfma",
                "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; see profiles/pmc_traffic.json "
                                  "for the calibration of FETCH_SIZE on this kernel's load shape" if traffic else None,
                "peak_measured": peak_meas, "avg_launch_ms": ms, "flops_per_unit": fl, "units_per_launch": nbl}
    # structured solvers: which stage dominates decides the kernel that is priced
    solve_ms, post_ms = stage["factor"] / K, stage["transform"] / K
    if solver == "flat":
        if post_ms >= solve_ms:
            by = bytes_fft_resid(N, M, T)
            ach = nbl * by / (post_ms * 1e-3) / 1e9
            return {"kernel": "k_fft_resid (back transform + model + residual + chi^2 + |z|^2 sums, one pass)",
                    "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": None, "avg_launch_ms": post_ms, "bytes_per_unit": by, "units_per_launch": nbl}
        by = bytes_flat(N, M, T)
        ach = nbl * by / (solve_ms * 1e-3) / 1e9
        return {"kernel": "k_solve_flat (diagonal + rank-M border system through its Schur complement)",
                "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": None, "avg_launch_ms": solve_ms, "bytes_per_unit": by, "flops_per_unit": flops_flat(N, M, T),
                "units_per_launch": nbl}
    fl = flops_lowrank(N, M, T, fmax)
    ach = nbl * fl / (solve_ms * 1e-3) / 1e12
    return {"kernel": "low-rank solve (k_flat_blocks, 3 x k_fft, k_lrf_gather, k_factor/k_backsolve of order M+f, "
                      "k_lrf_fill, k_lrf_back): the 'factor' stage as a whole",
            "bound": "mfma", "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": ach / FP64_MFMA_PEAK_TFLOPS, "traffic": None, "peak_measured": peak_meas,
            "avg_launch_ms": solve_ms, "flops_per_unit": fl, "units_per_launch": nbl,
            "note": "a chain of short HBM/latency-bound kernels; the flop model is the solver's own "
                    "(transforms + order-(M+f) Cholesky), not the dense factorisation's"}


def whole_step_for(solver, value_per_gpu, N, M, T, fmax):
    """Whole-iteration flop/byte model of the solver that ran (dense: SURVEY 8(d) F_alg / B_alg)."""
    if solver == "dense":
        fl, by = flops_unit(N, M, T), bytes_unit(N, M, T)
    else:
        post_fl = 2 * T * 5 * N * np.log2(N) + 8.0 * N * M * T
        post_by = 3 * 16.0 * T * N + 8.0 * N
        if solver == "flat":
            fl, by = flops_flat(N, M, T) + post_fl, bytes_flat(N, M, T) + post_by
        else:
            fl = flops_lowrank(N, M, T, fmax) + post_fl
            by = bytes_flat(N, M, T) + post_by + 2 * 16.0 * N * (1 + M + 2 * T) * 3
    return {"flops_per_unit": fl, "bytes_per_unit": by, "tflops": value_per_gpu * fl / 1e12,
            "frac_mfma": value_per_gpu * fl / 1e12 / FP64_MFMA_PEAK_TFLOPS,
            "frac_hbm": value_per_gpu * by / 1e9 / HBM_PEAK_GBS}


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script, one per GPU,
    and relay rank 0's JSON line.  The parent never imports torch, never loads libhpx and never
    touches a GPU (a process that has initialised HIP must not fork/exec GPU children); the
    children are plain `python bench.py ...` runs with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
    their environment, exactly what `torch.distributed.run` would have set.  Mirrors the
    reference's one-rank-per-block layout (run-hydra-pspec.py:268-287, :476-490)."""
    import socket
    import subprocess
    import tempfile
    n = args.gpus
    with socket.socket() as s:          # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + argv, env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=None))
    # wait for all ranks; if one fails the others would sit in a barrier for ever: end them
    codes = [None] * n
    deadline = time.time() + float(os.environ.get("HPX_BENCH_SPAWN_TIMEOUT", "3000"))
    while any(c is None for c in codes):
        for r, pr in enumerate(procs):
            if codes[r] is None:
                codes[r] = pr.poll()
        failed = any(c not in (None, 0) for c in codes) or time.time() > deadline
        if failed:
            for r, pr in enumerate(procs):     # exactly the children started above
                if codes[r] is None:
                    pr.terminate()
            for r, pr in enumerate(procs):
                if codes[r] is None:
                    try:
                        codes[r] = pr.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        pr.kill()
                        codes[r] = pr.wait()
            break
        time.sleep(0.05)
    out0.seek(0)
    sys.stdout.write(out0.read())
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


def init_ranks(args):
    """(rank, world, local_rank, dist-or-None, backend).  Refuses a launcher whose world size
    disagrees with --gpus."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    return rank, world, local_rank


def dry_run(args):
    """Rank plumbing without a GPU (tests/test_bench_spawn.py): every rank joins a gloo group,
    computes its block of baselines exactly as the real run does, takes part in the barrier and the
    max-over-ranks reduction, and rank 0 prints what every rank owns."""
    import torch
    import torch.distributed as dist
    from hydra_pspec_amd.sharding import split_counts
    rank, world, local_rank = init_ranks(args)
    if os.environ.get("HPX_BENCH_DRYRUN_FAIL_RANK") == str(rank):     # test hook: a rank that dies early
        sys.exit(3)
    nbl_gpu = args.nbl or CONFIGS[args.config][0]
    counts = split_counts(nbl_gpu * world, world)
    k0 = sum(counts[:rank])
    blocks = [(k0, k0 + counts[rank])]
    t = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.init_process_group("gloo")
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        got = [None] * world
        dist.all_gather_object(got, (rank, local_rank, k0, k0 + counts[rank], os.getpid()))
        blocks = [(g[2], g[3]) for g in sorted(got)]
        pids = [g[4] for g in sorted(got)]
        dist.barrier()
        dist.destroy_process_group()
    else:
        pids = [os.getpid()]
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "blocks": blocks, "pids": pids,
                          "baselines_total": int(sum(counts)), "max_time": float(t.item())}))


def bench_aux(args):
    """`--config dpss` / `--config oqe`: the two stand-alone library paths the north star names next to
    the Gibbs loop.  One step = one pass over the batch; durations from events on the launch stream."""
    import torch
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(int(os.environ.get("HPX_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    from hydra_pspec_amd import dpss, hpx
    K, W = args.steps, args.warmup
    dev = torch.device("cuda", torch.cuda.current_device())

    def timed(fn, n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n, e0.elapsed_time(e1) / n       # wall s, device ms per step

    if args.config == "dpss":
        ng, per, N, nm = args.nbl or 1024, 32, 512, 12
        rng = np.random.default_rng(1)
        freqs = np.linspace(100e6, 200e6, N)
        x = np.arange(N)
        cov = np.exp(-0.5 * ((x[:, None] - x[None, :]) / 3.0) ** 2) + 0.5 * np.eye(N)
        w = (rng.uniform(size=(ng, N)) > 0.15).astype(float)
        d_d = (torch.randn((ng, per, N), dtype=torch.float64, device=dev)
               + 1j * torch.randn((ng, per, N), dtype=torch.float64, device=dev)).contiguous()
        d_w = hpx.to_dev(torch, w, torch.float64, dev)
        pr = dpss.DpssProjector(ng, per, freqs, cov, nmodes=nm, alpha=6.0)
        out = torch.empty((ng, per, 2 * nm), dtype=torch.float64, device=dev)
        for _ in range(max(W, 1)):
            pr.fit(d_d, d_w, out=out)
        wall_full, ms_full = timed(lambda: pr.fit(d_d, d_w, out=out), K)
        wall_apply, ms_apply = timed(lambda: pr.fit(d_d, None, out=out), K)
        nspec = ng * per
        NP, ncol = N, 16
        by_apply = nspec * (N * 16.0 + 2 * nm * 8.0) + ng * NP * ncol * 16.0
        fl_group = ng * (8.0 * NP * NP * ncol + 8.0 * NP * ncol * nm)
        ms_group = max(ms_full - ms_apply, 1e-6)
        # CPU: the oracle's closed form on a bounded sample (the reference's L-BFGS-B fit takes ~1 s per spectrum)
        from oracle import dpss_ref
        hd, t0 = d_d[:2, :4].cpu().numpy(), time.perf_counter()
        worst = 0.0
        ho = out[:2, :4].cpu().numpy()
        for g in range(2):
            for t in range(4):
                _, cf = dpss_ref.dpss_fit_closed_form(hd[g, t], w[g], freqs, cov, nmodes=nm, alpha=6.0)
                worst = max(worst, float(np.max(np.abs(ho[g, t] - cf)) / np.max(np.abs(cf))))
        cpu_s = (time.perf_counter() - t0) / 8
        res = {"metric": "DPSS foreground fits (spectra) per second at Nfreq=512, 12 modes", "value": nspec / wall_full,
               "unit": "spectra/s", "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": wall_full * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"dpss: {ng} baselines x {per} times, Nfreq {N}, {nm} DPSS modes, 15 % flagged "
                                      "channels per baseline (weights shared by the times of a baseline), smooth + "
                                      "white covariance; hpx_dpss_project_grouped with a caller workspace"},
               "value_projection_only": nspec / wall_apply,
               "roofline": {"kernel": "k_dpss_apply (tall-skinny projection P d + nm x nm multiply, f64 MFMA, visibility "
                                      "cube read once through LDS tiles)", "bound": "hbm",
                            "achieved": by_apply / (ms_apply * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": by_apply / (ms_apply * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                            "avg_launch_ms": ms_apply, "bytes_per_unit": by_apply / nspec, "units_per_launch": nspec},
               "group_stage": {"kernel": "k_dft (dense product inv(cov) x weighted modes, f64 MFMA) + k_dpss_group "
                                         "(normal matrix, inverse, projector), once per set of weights",
                               "bound": "mfma", "achieved": fl_group / (ms_group * 1e-3) / 1e12,
                               "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": fl_group / (ms_group * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                               "avg_ms": ms_group, "flops_per_group": fl_group / ng},
               "cpu_baseline": {"value": 1.0 / cpu_s, "unit": "spectra/s", "cores": 1, "kind": "port",
                                "sample": "oracle closed form (numpy normal equations incl. inv(cov)) on 8 spectra; "
                                          "the reference's own L-BFGS-B fit is ~100x slower per spectrum"},
               "max_rel_dev_vs_cpu": worst}
        print(json.dumps(res))
        return

    if args.config == "fgmodes":
        # leading eigenvectors of np.cov(bl_data.T) per baseline (scripts/calc-vis-cov-matrices.py:235-249)
        from hydra_pspec_amd import fgmodes
        nb, T, N, nm = args.nbl or 1024, 32, 512, 12
        if args.order:
            N, T = args.order, args.order + 8
        rng = np.random.default_rng(2)
        nu = np.linspace(-1, 1, N)
        basis = np.stack([np.cos(np.pi * k * nu / 2 + 0.3 * k) * np.exp(0.2j * k * nu) for k in range(16)], axis=1)
        nsrc = min(nb, 64)       # distinct cubes (the order-512 case would otherwise spend minutes in the host RNG)
        amps = (rng.standard_normal((nsrc, T, 16)) + 1j * rng.standard_normal((nsrc, T, 16))) * (2.0 ** -np.arange(16))
        vis = amps @ basis.T + 0.01 * (rng.standard_normal((nsrc, T, N)) + 1j * rng.standard_normal((nsrc, T, N)))
        d_vis = hpx.to_dev(torch, vis, torch.complex128, dev)
        if nsrc < nb:
            d_vis = d_vis.repeat((nb + nsrc - 1) // nsrc, 1, 1)[:nb].contiguous()
        out = {}

        def run():
            out["m"], out["e"] = fgmodes.cov_eig_modes(d_vis, nm, return_evals=True, as_numpy=False)
        for _ in range(max(W, 1)):
            run()
        wall, ms = timed(run, K)
        t0 = time.perf_counter()
        worst = 0.0
        hm, he = out["m"][:4].cpu().numpy(), out["e"][:4].cpu().numpy()
        for b in range(4):
            lam, U = np.linalg.eigh(np.cov(vis[b % nsrc].T))
            lam, U = lam[::-1][:nm], U[:, ::-1][:, :nm]
            worst = max(worst, float(np.max(np.abs(he[b] / lam - 1))),
                        float(np.max(1 - np.abs(np.sum(U.conj() * hm[b], axis=0)))))
        cpu_s = (time.perf_counter() - t0) / 4
        by = nb * (16.0 * T * N + 16.0 * N * nm + 8.0 * nm)
        if args.order:
            # flop model of the blocked one-sided Jacobi (csrc/hpx_eigh.hip): per visit of a pair of 8-column blocks the
            # 16 x 16 Gram matrix and the update W Q, 8 * 16 * 16 * n flops each; (n/16)(n/8 - 1) visits per sweep;
            # + the covariance (8 T n^2) and its Cholesky factor (4/3 n^3)
            n_ = ((N + 15) // 16) * 16
            sweeps = int(os.environ.get("HPX_BENCH_EIGH_SWEEPS", "10"))
            fl = sweeps * (n_ // 16) * (n_ // 8 - 1) * 2 * 8 * 256 * n_ + 8.0 * T * N * N + 4.0 / 3.0 * n_ ** 3
            ach = nb * fl / (ms * 1e-3) / 1e12
            res = {"metric": f"foreground-mode sets (baselines) per second: leading eigenvectors of np.cov(vis.T), "
                             f"Ntimes {T}, Nfreq {N}, {nm} modes (the Nfreq x Nfreq covariance is diagonalised)",
                   "value": nb / wall, "unit": "baselines/s", "n_gpus": 1, "steps": K, "warmup": W,
                   "ms_per_step": wall * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                   "dtype": "f64", "data": "synthetic",
                   "config": {"workload": f"fgmodes --order {N}: {nb} baselines x (Ntimes {T}, Nfreq {N}), {nm} modes; "
                                          "hpx_fgmodes_eig: centring, covariance, Cholesky factor, blocked one-sided "
                                          "Jacobi (hpx_eigh.hip), mode selection; includes its workspace allocation"},
                   "roofline": {"kernel": "k_hj_step (16 x 16 block rotations of the one-sided Jacobi on the FP64 MFMA)",
                                "bound": "mfma", "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": ach / FP64_MFMA_PEAK_TFLOPS, "traffic": None, "avg_launch_ms": ms,
                                "flops_per_unit": fl, "units_per_launch": nb,
                                "note": f"flop model with {sweeps} sweeps (HPX_EIGH_TRACE=1 prints the count); the "
                                        "method is bound by HBM: every step streams the factor once in and once out "
                                        f"({sweeps} x {n_ // 8 - 1} steps x {2 * 16 * n_ * n_ / 1e6:.1f} MB per baseline)"},
                   "cpu_baseline": {"value": 1.0 / cpu_s, "unit": "baselines/s", "cores": os.cpu_count(), "kind": "port",
                                    "sample": "np.cov + np.linalg.eigh (the reference script's own calls, default BLAS "
                                              "threads) on 4 baselines, incl. the comparison"},
                   "max_rel_dev_vs_cpu": worst}
            print(json.dumps(res))
            return
        res = {"metric": "foreground-mode sets (baselines) per second: leading eigenvectors of np.cov(vis.T), "
                         "Ntimes 32, Nfreq 512, 12 modes", "value": nb / wall, "unit": "baselines/s", "n_gpus": 1,
               "steps": K, "warmup": W, "ms_per_step": wall * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"fgmodes: {nb} baselines x (Ntimes {T}, Nfreq {N}), {nm} modes through the "
                                      "Ntimes x Ntimes Gram matrix (hpx_fgmodes_eig: centring, Gram matrix, cyclic "
                                      "Jacobi, mode reconstruction); includes the allocation of its workspace"},
               "roofline": {"kernel": "hpx_fgmodes_eig (k_center + k_gram + k_jacobi + k_modes_out)", "bound": "hbm",
                            "achieved": by / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": ms,
                            "bytes_per_unit": by / nb, "units_per_launch": nb,
                            "note": "algorithmic bytes = the visibility cube in, the modes and eigenvalues out"},
               "cpu_baseline": {"value": 1.0 / cpu_s, "unit": "baselines/s", "cores": os.cpu_count(), "kind": "port",
                                "sample": "np.cov + np.linalg.eigh (the reference script's own calls, default BLAS "
                                          "threads) on 4 baselines, incl. the comparison"},
               "max_rel_dev_vs_cpu": worst}
        print(json.dumps(res))
        return

    # ---- oqe
    nb, s = args.nbl or 64, 512
    # Hermitian, diagonally dominant weightings (input synthesis only: element-wise, no library GEMM anywhere)
    a = (torch.randn((nb, s, s), dtype=torch.float64, device=dev) + 1j * torch.randn((nb, s, s), dtype=torch.float64, device=dev)) / s
    R = (0.5 * (a + a.conj().transpose(1, 2)) + 2.0 * torch.eye(s, dtype=torch.complex128, device=dev)).contiguous()
    Fo = torch.empty_like(R)
    L = hpx.lib()
    nbytes = int(L.hpx_oqe_workspace_bytes(nb, s, 0))
    work = torch.empty(nbytes // 8 + 1, dtype=torch.float64, device=dev)

    def fisher():
        hpx.check(L.hpx_oqe_fisher(nb, s, hpx.ptr(R), hpx.ptr(Fo), 0, hpx.ptr(work), nbytes, hpx.stream_ptr(torch)),
                  "hpx_oqe_fisher")
    for _ in range(max(W, 1)):
        fisher()
    wall, ms = timed(fisher, K)
    SP = s
    fl = nb * 4 * 8.0 * SP ** 3
    from oracle import oqe_ref
    R0 = R[0].cpu().numpy()
    t0 = time.perf_counter()
    Fc = oqe_ref.F_closed(s, R0)
    cpu_s = time.perf_counter() - t0
    dev_err = float(np.max(np.abs(Fo[0].cpu().numpy() - Fc)) / np.max(np.abs(Fc)))
    res = {"metric": "OQE Fisher matrices per second at s=512 (F_ab = 1/2 tr(R* Q_a R Q_b), general R)",
           "value": nb / wall, "unit": "matrices/s", "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": wall * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"oqe: {nb} weightings R of order {s} -> Fisher matrices (hpx_oqe_fisher, variant 0, "
                                  "caller workspace): four dense s^3 products with the DFT matrix on the f64 MFMA"},
           "roofline": {"kernel": "k_dft (dense complex s x s x s products M R, (.) M^H, conj(M) R, (.) M^T)",
                        "bound": "mfma", "achieved": fl / (ms * 1e-3) / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": fl / (ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
                        "avg_launch_ms": ms, "flops_per_unit": fl / nb, "units_per_launch": nb},
           "cpu_baseline": {"value": 1.0 / cpu_s, "unit": "matrices/s", "cores": os.cpu_count(), "kind": "port",
                            "sample": "oracle closed form (two numpy s^3 products per side, default BLAS threads) on 1 "
                                      "matrix; the reference's own O(s^5) trace loops cannot run at s = 512"},
           "max_rel_dev_vs_cpu": dev_err}
    print(json.dumps(res))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C3", choices=sorted(CONFIGS) + ["dpss", "oqe", "fgmodes"])
    ap.add_argument("--nbl", type=int, default=None, help="baselines per GPU (default: config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-length", action="store_true",
                    help="skip the full-length leg (the config's 1000 / 2000 iterations, ~10 s at C3)")
    ap.add_argument("--solver", default="dense", choices=["dense", "auto"],
                    help="dense (default): the general batched-Cholesky path the metric is about; auto: let "
                         "unflagged flat-noise batches take the structured solve (reported separately anyway)")
    ap.add_argument("--noise", default="diag", choices=["diag", "dense", "pertime-dense"],
                    help="diag (default): the configs' diagonal inverse noise variances.  dense: a Hermitian banded "
                         "inverse noise covariance per baseline (correlated noise; with --flag-frac > 0 the Woodbury "
                         "path).  pertime-dense: one such matrix per (baseline, time) with time-dependent flags "
                         "(needs --nbl small: nbl x Ntimes full matrices)")
    ap.add_argument("--flag-frac", type=float, default=None, help="override the config's flag fraction")
    ap.add_argument("--order", type=int, default=0,
                    help="--config fgmodes: diagonalise the Nfreq x Nfreq covariance itself (Ntimes = order + 8 > "
                         "Nfreq = order, scripts/calc-vis-cov-matrices.py:239-247) instead of the Ntimes x Ntimes Gram matrix")
    ap.add_argument("--dry-run", action="store_true",
                    help="rank plumbing only (gloo, no GPU): print the blocks of baselines the ranks would own")
    args = ap.parse_args()

    # `python bench.py --gpus N` on its own: this process becomes a launcher and nothing else
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    if args.dry_run:
        return dry_run(args)
    if args.config in ("dpss", "oqe", "fgmodes"):
        return bench_aux(args)

    rank, world, local_rank = init_ranks(args)
    import torch
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # rehearsal knobs (one-GPU box): HPX_BENCH_DEVICE pins every rank to one device and
    # HPX_BENCH_BACKEND=gloo replaces RCCL for the timing barrier / max-reduction
    dev_index = int(os.environ.get("HPX_BENCH_DEVICE", local_rank))
    backend = os.environ.get("HPX_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    from hydra_pspec_amd import hpx, pspec, synthetic
    from hydra_pspec_amd.sharding import split_counts
    nbl_gpu, T, N, M, frac = CONFIGS[args.config]
    if args.flag_frac is not None:
        frac = args.flag_frac
    if args.nbl:
        nbl_gpu = args.nbl
    elif args.noise == "pertime-dense":
        nbl_gpu = 8            # nbl x Ntimes matrices of N x N: 8 x 32 x 4 MiB at N = 512
    counts = split_counts(nbl_gpu * world, world)
    k0 = sum(counts[:rank])
    nbl = counts[rank]
    K, W = args.steps, args.warmup

    d = synthetic.make_baselines(N, T, M, k0=k0, nbl=nbl, flag_frac=frac, dense=False)
    flags_in, ninv_in = d["flags"], d["ninv_diag"]
    if args.config == "N4":
        rng = np.random.default_rng(11 + k0)
        flags_in = np.broadcast_to(d["flags"][:, None, :], (nbl, T, N)).copy()
        flags_in &= rng.uniform(size=(nbl, T, N)) > 0.05
        ninv_in = np.ascontiguousarray(np.broadcast_to(d["ninv_diag"][:, None, :] *
                                                       rng.uniform(0.7, 1.3, size=(nbl, T, 1)), (nbl, T, N)))
    ninv_dense = None
    if args.noise != "diag":
        # Hermitian banded inverse noise covariance, one per baseline (scaled to the baseline's noise level), as
        # tests/test_gpu_fullsize.py:_banded_ninv; pertime-dense: one per (baseline, time) with 5 % more flags per time
        i = np.arange(N)
        band = np.zeros((N, N), dtype=complex)
        band[i, i] = 1.0 + 0.2 * np.cos(0.3 * i)
        band[i[:-1], i[:-1] + 1] = 0.3 * np.exp(0.4j)
        band[i[:-1] + 1, i[:-1]] = 0.3 * np.exp(-0.4j)
        band[i[:-2], i[:-2] + 2] = 0.1
        band[i[:-2] + 2, i[:-2]] = 0.1
        binv = np.linalg.inv(band)
        lev = np.asarray(d["ninv_diag"])[:, 0]
        if args.noise == "dense":
            ninv_dense = np.ascontiguousarray(binv[None] * lev[:, None, None])
        else:
            rng = np.random.default_rng(11 + k0)
            flags_in = np.broadcast_to(d["flags"][:, None, :], (nbl, T, N)).copy()
            flags_in &= rng.uniform(size=(nbl, T, N)) > 0.05
            ninv_dense = np.ascontiguousarray(binv[None, None] * (lev[:, None, None, None] *
                                                                  rng.uniform(0.7, 1.3, size=(nbl, T, 1, 1))))
        ninv_in = None
    torch.cuda.synchronize()
    t_setup = time.perf_counter()
    # ranks that share a GPU (a launcher rehearsal) must not take the split factorisation: its co-operating
    # workgroups need the device to themselves (hpx.h, HPX_OPT_FACTOR_SPLIT)
    shared_gpu = world > 1 and "HPX_BENCH_DEVICE" in os.environ
    gb = pspec.GibbsBatch(d["vis"], flags_in, d["fgmodes"], ninv_in, d["ps_prior"],
                          W + K, seed=d["seed"], solver=args.solver, ninv_dense=ninv_dense,
                          allow_split=not shared_gpu)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup
    ps0 = np.broadcast_to(d["ps0"], (nbl, N)).copy()
    fmax = int((~np.asarray(d["flags"]).astype(bool)).sum(axis=1).max())
    per_time = args.config == "N4" or args.noise == "pertime-dense"
    units_per_bl = T if per_time else 1      # factorisations per baseline and iteration

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if W > 0:
        first = gb.run(W, ps0=ps0)
    # ---- the timed region (the metric): exactly K steps, no profiling events, barrier + synchronize on both sides
    barrier()
    gb.plan.set_profiling(False)
    t0 = time.perf_counter()
    out = gb.run(K, ps0=ps0 if W == 0 else None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        dist.barrier()
    torch.cuda.synchronize()
    # ---- the same K steps again with a HIP event between the stages of every iteration (hpx_plan_stage_ms):
    # where `stage_ms_per_step` and `roofline.avg_launch_ms` come from; ~2 % slower than the metric's run
    gb.iter_done = W
    barrier()
    gb.plan.set_profiling(True)
    t0 = time.perf_counter()
    out_ev = gb.run(K, ps0=(first["ps_last"] if W > 0 else ps0))
    torch.cuda.synchronize()
    dt_events = time.perf_counter() - t0
    stage = gb.plan.stage_ms()
    events_same_chain = bool(torch.equal(out_ev["signal_ps"], out["signal_ps"]))
    del out_ev
    if dist is not None:
        tt = torch.tensor([dt_events], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt_events = float(tt.item())
        dist.barrier()
    torch.cuda.synchronize()

    # SURVEY 8(d)'s inclusive variant of the metric: the same K steps again, this time with the
    # random tables uploaded from host memory and P(k) / ln-posterior downloaded inside the clock
    gb.plan.set_profiling(False)
    uni_h, igy_h = pspec.draw_tables(T, N, W + K, d["seed"])
    ps_host = torch.empty((nbl, K, N), dtype=torch.float64, pin_memory=True)
    ln_host = torch.empty((nbl, K), dtype=torch.float64, pin_memory=True)
    gb.iter_done = W
    barrier()
    t1 = time.perf_counter()
    gb.set_tables(uni_h, igy_h)
    o2 = gb.run(K, ps0=(first["ps_last"] if W > 0 else ps0))
    ps_host.copy_(o2["signal_ps"], non_blocking=True)
    ln_host.copy_(o2["ln_post"], non_blocking=True)
    torch.cuda.synchronize()
    dt_incl = time.perf_counter() - t1
    same_chain = bool(torch.equal(o2["signal_ps"], out["signal_ps"]))
    if dist is not None:
        tt = torch.tensor([dt_incl], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt_incl = float(tt.item())
    del o2

    # the config as BASELINE.json states it: the same batch for the config's full iteration count (its own
    # plan and tables; bracketed like the timed region, max over ranks)
    full_len = None
    n_full = FULL_ITERS.get(args.config)
    special = args.noise != "diag"     # (the extra legs below are the diagonal-noise metric's)
    if n_full and not args.no_full_length and not special:
        gfull = pspec.GibbsBatch(d["vis"], flags_in, d["fgmodes"], ninv_in, d["ps_prior"], n_full, seed=d["seed"],
                                 solver=args.solver)
        barrier()
        t1 = time.perf_counter()
        ofull = gfull.run(n_full, ps0=ps0)
        torch.cuda.synchronize()
        dt_full = time.perf_counter() - t1
        finite = bool(torch.isfinite(ofull["signal_ps"]).all()) and bool(torch.isfinite(ofull["ln_post"]).all())
        prefix_same = bool(torch.equal(ofull["signal_ps"][:, W:W + K], out["signal_ps"])) if W + K <= n_full else None
        if dist is not None:
            tt = torch.tensor([dt_full], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_full = float(tt.item())
        full_len = {"iters": n_full, "seconds": dt_full, "value": sum(counts) * n_full / dt_full,
                    "unit": "baseline*iter/s", "all_finite": finite, "timed_steps_are_its_iterations": prefix_same,
                    "note": "the same batch for the iteration count BASELINE.json names for this config, one "
                            "hpx_gibbs_run call, no profiling events; inputs resident, P(k) history kept on the device"}
        del ofull
        gfull.close()

    # the same batch through solver="auto" (outside the timed region): unflagged flat-noise inputs
    # such as C3's then take the O(N M (M+T)) structured solve instead of the dense factorisation
    flat_extra = None
    if rank == 0 and world == 1 and args.solver == "dense" and args.config != "N4" and not special:
        gf = pspec.GibbsBatch(d["vis"], d["flags"], d["fgmodes"], d["ninv_diag"], d["ps_prior"],
                              W + K, seed=d["seed"], solver="auto")
        if gf.solver in ("flat", "lowrank"):
            fo = gf.run(W, ps0=ps0) if W > 0 else None
            torch.cuda.synchronize()
            gf.plan.set_profiling(True)
            t1 = time.perf_counter()
            fo = gf.run(K, ps0=ps0 if W == 0 else None)
            torch.cuda.synchronize()
            dtf = time.perf_counter() - t1
            dev = float((fo["signal_ps"] / out["signal_ps"] - 1).abs().max().item())
            fstage = gf.plan.stage_ms()
            flat_extra = {"value": nbl * K / dtf, "unit": "baseline*iter/s", "ms_per_step": dtf / K * 1e3,
                          "stage_ms_per_step": {k: v / K for k, v in fstage.items()},
                          "pk_max_rel_dev_vs_dense": dev,
                          "solver": gf.solver,
                          "roofline": roofline_for(gf.solver, fstage, nbl, N, M, T, fmax, K, None, None),
                          "whole_step": whole_step_for(gf.solver, nbl * K / dtf, N, M, T, fmax),
                          "note": "solver='auto' on the same batch: one Ninv value over the unflagged channels -> "
                                  "diagonal + border system solved through its Schur complement (hpx_flat.hip "
                                  "without flags, hpx_lowrank.hip with flags); not the headline value, which stays "
                                  "on the general dense path"}
        gf.close()

    # which physical device every rank ran on (a rehearsal pins all ranks to one: HPX_BENCH_DEVICE).  The
    # identity is (host name, device uuid / PCI bus id), not the local index: ranks on two nodes, or ranks
    # that each see their card as device 0 through HIP_VISIBLE_DEVICES, must not collapse into one device.
    import hashlib, socket
    prop = torch.cuda.get_device_properties(dev_index)
    ident = f"{socket.gethostname()}|{getattr(prop, 'uuid', '')}|{getattr(prop, 'pci_bus_id', '')}|" \
            f"{getattr(prop, 'pci_device_id', '')}|{os.environ.get('HIP_VISIBLE_DEVICES', '')}|{dev_index}"
    my_id = int.from_bytes(hashlib.sha1(ident.encode()).digest()[:6], "big")      # 48 bits: exact in a float64
    devices = [my_id]
    if dist is not None:      # every rank's id in its own slot, summed over the ranks (a plain all-reduce, as for the times)
        ids = torch.zeros(world, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        ids[rank] = float(my_id)
        dist.all_reduce(ids, op=dist.ReduceOp.SUM)
        devices = [int(v) for v in ids.tolist()]
    if rank == 0:
        total_units = sum(counts) * K
        value = total_units / dt
        traffic, step_traffic, traffic_stale = None, None, False
        if gb.solver == "dense":
            try:   # HBM bytes per k_factor launch from the committed PMC passes (profiles/), same workload --
                # only when they were measured on THIS library's sources
                pmall = json.load(open(REPO / "profiles" / "pmc_traffic.json"))
                pmj = pmall[args.config]
                if pmall.get("source_hash") == kernel_source_hash():
                    pm = pmj["k_factor"]
                    traffic = pm["bytes_per_launch"] * nbl / pm["baselines"]
                    if "step" in pmj:
                        step_traffic = pmj["step"]["bytes_per_step"] * nbl / pmj["step"]["baselines"]
                else:
                    traffic_stale = True
            except Exception:
                pass
        peak_meas = np.zeros(1)
        import ctypes
        hpx.check(hpx.lib().hpx_mfma_f64_peak(20000, peak_meas.ctypes.data_as(ctypes.c_void_p)))
        wb_cols = fmax if (special and fmax > 0) else 0      # Woodbury columns next to the data columns (dense noise + flags)
        if args.noise == "pertime-dense":
            wb_cols = int((~flags_in).sum(axis=2).max()) if not flags_in.all() else 0
        if per_time:      # nbl*T systems with one right-hand side each (+ the Woodbury columns)
            roof = roofline_for("dense", stage, nbl * T, N, M, 1 + wb_cols, fmax, K, None, float(peak_meas[0]))
            roof["whole_step"] = whole_step_for("dense", value / world * T, N, M, 1 + wb_cols, fmax)
            roof["note"] = ("time-dependent flags: one (N+M)-order system per (baseline, time) and iteration; "
                            f"units_per_launch = {nbl} baselines x {T} times"
                            + (f"; {wb_cols} Woodbury right-hand sides per unit" if wb_cols else ""))
        elif special:
            roof = roofline_for("dense", stage, nbl, N, M, T + wb_cols, fmax, K, None, float(peak_meas[0]))
            roof["whole_step"] = whole_step_for("dense", value / world, N, M, T + wb_cols, fmax)
            roof["note"] = (f"Hermitian non-diagonal Ninv: the matrix is assembled in full every iteration (k_assemble), "
                            f"{T} data columns + {wb_cols} Woodbury columns (one per flagged channel) go through the "
                            "factorisation's forward solve")
        else:
            roof = roofline_for(gb.solver, stage, nbl, N, M, T, fmax, K, traffic, float(peak_meas[0]))
            roof["whole_step"] = whole_step_for(gb.solver, value / world, N, M, T, fmax)
            if step_traffic:     # measured HBM-side bytes of one whole iteration (PMC passes) against this run's step time
                gbs = step_traffic / (dt / K) / 1e9
                roof["whole_step"].update(traffic_measured=step_traffic, hbm_gbs_measured=gbs,
                                          frac_hbm_measured=gbs / HBM_PEAK_GBS,
                                          frac_hbm_achievable=gbs / 6300.0)
        res = {
            "metric": "baseline x Gibbs-iter/sec at Nfreq=512; P(k) rtol vs CPU ref",
            "value": value, "unit": "baseline*iter/s", "n_gpus": len(set(devices)), "steps": K, "warmup": W,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.config}: {nbl_gpu} synthetic baselines per GPU x "
                                   f"(Ntimes {T}, Nfreq {N}, {M} DPSS fg modes), flag fraction {frac}, "
                                   "7-bin prior, fp64",
                       "baselines_total": int(sum(counts)), "sharding": "contiguous blocks by baseline index, "
                       "no collective"},
            "value_no_events": value,
            "value_with_stage_events": total_units / dt_events,
            "events_note": "`value` is the event-free run (exactly K steps between barrier + synchronize); "
                           "`value_with_stage_events` is the same K steps again with a HIP event between the stages of "
                           "every iteration -- the run `stage_ms_per_step` and `roofline.avg_launch_ms` come from; same "
                           f"chain bit for bit: {events_same_chain}",
            "value_incl_transfers": total_units / dt_incl,
            "incl_transfers_note": f"the same {K} steps with the (Niter, Nfreq) random tables uploaded from host memory "
                                   f"and P(k) + ln-posterior ({(ps_host.numel() + ln_host.numel()) * 8 / 1e6:.0f} MB per GPU) "
                                   "copied to pinned host memory inside the clock (SURVEY 8d); same chain bit for bit: "
                                   f"{same_chain}",
            "roofline": roof,
            "stage_ms_per_step": {k: v / K for k, v in stage.items()},
        }
        res["config"]["solver"] = gb.solver
        if len(set(devices)) != world or "HPX_BENCH_BACKEND" in os.environ:
            res["rehearsal"] = {"ranks": world, "ranks_per_device": world // len(set(devices)), "backend": backend,
                                "note": "launcher rehearsal: several ranks share one physical GPU -- NOT a multi-GPU result"}
            res["scaling"] = "rehearsal"
        if traffic_stale:
            roof["traffic_note"] = ("profiles/pmc_traffic.json was measured on other kernel sources (source_hash "
                                    "differs): no traffic figure for this build")
        if full_len:
            res["full_length"] = full_len
        if flat_extra:
            res["flat_noise_structured_solve"] = flat_extra     # (key kept from the unflagged case)
        if per_time:
            res["config"]["workload"] += (f"; TIME-DEPENDENT flags (5 % per time on top) and noise levels: "
                                          f"{nbl_gpu * T} factorisations per iteration")
            res["systems_per_second"] = value * units_per_bl
        res["setup_seconds"] = t_setup
        if special:
            res["config"]["noise"] = ("a Hermitian banded inverse noise covariance per baseline" if args.noise == "dense" else
                                      "a Hermitian banded inverse noise covariance per (baseline, time)")
            res["setup_note"] = ("setup_seconds = GibbsBatch(...): upload, the noise matrices' square roots "
                                 "(hpx_sqrtm_hpd_batched on the device), invariant operators; outside the timed region")
        if world == 1 and not args.no_cpu_baseline and args.config != "N4" and not special:
            one, ref_ps, dd = cpu_baseline(N, T, M, frac)
            mp = cpu_baseline_multiproc(N, T, M, frac)
            if mp and mp["value"] > one["value"]:
                # the stronger CPU configuration is the baseline: P single-thread processes, one
                # baseline stream each (the reference's MPI model); the one-process run rides along
                cb = dict(value=mp["value"], unit=mp["unit"], cores=mp["processes"], kind="port",
                          sample=mp["sample"], single_process=one)
            else:
                cb = dict(one, multi_process=mp)
            cb["note"] = ("the port drops the reference's per-iteration multiprocess.Pool fork and is ~1.7x faster "
                          "than the reference itself per iteration (0.93 vs 1.60 s at (32, 512, 12) on the build "
                          "container): a conservative baseline")
            res["cpu_baseline"] = cb
            # P(k) deviation of the GPU chain from the CPU chain on the same baselines/seed
            chk = pspec.gibbs_sample_with_fg_batched(dd["vis"], dd["flags"], dd["fgmodes"], dd["ninv_diag"],
                                                     dd["ps_prior"], ps_initial=dd["ps0"],
                                                     Niter=ref_ps.shape[1], seed=dd["seed"], solver=args.solver)
            res["pk_max_rel_dev_vs_cpu"] = float(np.max(np.abs(chk["signal_ps"] / ref_ps - 1)))
            res["speedup_vs_cpu_baseline"] = value / cb["value"]
        print(json.dumps(res))
    gb.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
