#!/usr/bin/env python3
"""Driver for the MI355X Gibbs path with the reference driver's call surface.

Counterpart of the reference's ``run-hydra-pspec.py`` for the hot path only
(SURVEY 8b "driver counterpart"): same flag / YAML key names and defaults
(reference run-hydra-pspec.py:39-239), same per-baseline prior construction
(:509-517), ``w = ~flags`` and the any-time-flagged channel mask (:493, :531-535),
same output tree ``<out_dir>/<dirname>/<ant1>-<ant2>/{gcr-eor,cov-eor,dps-eor,
fg-amps,chisq,ln-post}.npy`` (:498-502, utils.py:307-312) and ``timings.json`` keys
(:570-581).  What differs, on purpose:

* no MPI: one process per GPU (``torchrun`` / RANK, WORLD_SIZE, LOCAL_RANK); baselines
  are split over ranks with the reference's block rule (:268-287) and every rank reads
  only its own block -- no scatter/gather;
* all baselines of a rank run as ONE batch on the GPU;
* inputs: ``.npy``/``.npz`` visibility cubes (``--file_paths cube.npy`` with shape
  (Nbl,Ntimes,Nfreqs), optional ``antpairs.npy``), or ``--synthetic Nbl,Ntimes,Nfreqs``;
  UVH5 files are read by the package's own HDF5 reader (hydra_pspec_amd/uvh5.py: no
  pyuvdata, h5py or astropy), each rank only its own baselines, with the reference's
  frequency selection, ant1<ant2 conjugation and XX+YY pseudo-Stokes I.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path
from resource import RUSAGE_SELF, getrusage

import numpy as np
import yaml

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))


def build_parser():
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--config", type=str, help="YAML file whose keys are the flag names below")
    p.add_argument("--file_paths", type=str, nargs="+", help=".npy/.npz cube(s) or UVH5 file(s)")
    p.add_argument("--synthetic", type=str, help="Nbl,Ntimes,Nfreqs: synthetic inputs (SURVEY 8d recipe)")
    p.add_argument("--ant_str", type=str, default="cross")
    p.add_argument("--sigcov0", type=str)
    p.add_argument("--sigcov0_file", type=str)
    p.add_argument("--Nfgmodes", type=int, default=8)
    p.add_argument("--fgmodes", type=str)
    p.add_argument("--fgmodes_file", type=str)
    p.add_argument("--freq_range", type=str)
    p.add_argument("--flags", type=str)
    p.add_argument("--flags_file", type=str)
    p.add_argument("--noise", type=str)
    p.add_argument("--noise_file", type=str)
    p.add_argument("--noise_cov", type=str)
    p.add_argument("--noise_cov_file", type=str)
    p.add_argument("--nsamples", type=str)
    p.add_argument("--nsamples_file", type=str)
    p.add_argument("--n_ps_prior_bins", type=int, default=3)
    p.add_argument("--ps_prior_lo", type=float, default=0.0)
    p.add_argument("--ps_prior_hi", type=float, default=0.0)
    p.add_argument("--map_estimate", action="store_true")
    p.add_argument("--Niter", type=int, default=100)
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("-v", "--verbose", action="store_true")
    p.add_argument("--Nproc", type=int, default=1, help="accepted and ignored")
    p.add_argument("--out_dir", type=str, default="./")
    p.add_argument("--dirname", type=str, default=None)
    p.add_argument("--clobber", action="store_true")
    p.add_argument("--write_Niter", type=int, default=100)
    p.add_argument("--resume", action="store_true",
                   help="continue chains from the checkpoints in the output tree (written every --write_Niter "
                        "iterations); the earlier run's args.json must name the same inputs and seed")
    p.add_argument("--per_time_flags", action="store_true",
                   help="keep the flags time dependent: every time sample is solved with its own noise matrix "
                        "(Nbl x Ntimes factorisations per iteration) instead of masking a channel at all times "
                        "as soon as one sample is flagged, which is what the reference driver does (:524-541)")
    p.add_argument("--outputs", type=str, default="all",
                   help="'all' (the six reference files) or 'ps' (dps-eor.npy and ln-post.npy only)")
    p.add_argument("--solver", type=str, default="auto", choices=["auto", "dense"],
                   help="'auto': baselines whose unflagged channels share one noise variance take the structured exact "
                        "solve, everything else the batched dense Cholesky; 'dense': the dense Cholesky for all")
    p.add_argument("--thin", type=int, default=1,
                   help="keep every K-th iteration of the large histories (gcr-eor.npy, fg-amps.npy, chisq.npy: rows "
                        "0, K, 2K, ...); dps-eor.npy and ln-post.npy always hold every iteration.  --write_Niter must be "
                        "a multiple of K")
    p.add_argument("--host_mem_gb", type=float, default=None,
                   help="budget for the two pinned staging buffers of the output drain (default: half of MemAvailable); "
                        "a run whose chunks do not fit stops BEFORE sampling, with the numbers")
    p.add_argument("--fsync", action="store_true",
                   help="force every flush to disk (data, then the header) before the next one: checkpoints survive "
                        "a power loss, not only a crash of the process")
    p.add_argument("--write_workers", type=int, default=4, help="file-writing threads of the output drain")
    p.add_argument("--dry_run", action="store_true",
                   help="everything but the sampling (no GPU touched): inputs, rank blocks, output tree, "
                        "checkpoint files (zeros) and timings -- exercises the multi-rank plumbing")
    return p


def parse_args(argv=None):
    parser = build_parser()
    args = parser.parse_args(argv)
    if args.config:
        with open(args.config) as f:
            cfg = yaml.safe_load(f) or {}
        known = {a.dest for a in parser._actions}
        for k, v in cfg.items():
            if k not in known:
                raise SystemExit(f"unknown config key: {k}")
            if getattr(args, k) == parser.get_default(k):
                setattr(args, k, v)
    return args


def make_ps_prior(Nfreqs, n_bins, lo, hi):
    """(2,Nfreqs) rows [hi, lo] on Nfreqs//2 +- n_bins (run-hydra-pspec.py:509-517)."""
    pr = np.zeros((2, Nfreqs))
    if lo != 0 or hi != 0:
        sl = slice(Nfreqs // 2 - n_bins, Nfreqs // 2 + n_bins + 1)
        pr[0, sl] = hi
        pr[1, sl] = lo
    return pr


def any_time_unflagged(w):
    """Channel mask: True where NO time sample is flagged (run-hydra-pspec.py:531-535);
    ``w`` (Ntimes,Nfreqs) bool, True = good."""
    return np.all(w, axis=0)


def _parent_start_ticks():
    """Start time of the parent process (clock ticks since boot, /proc/<ppid>/stat field 22): with its pid it names
    ONE launch of plain `RANK=.. WORLD_SIZE=..` processes started by one shell."""
    try:
        with open(f"/proc/{os.getppid()}/stat") as f:
            return f.read().rsplit(")", 1)[1].split()[19]
    except (OSError, IndexError):
        return "0"


def launch_token():
    """What the ranks of ONE launch have in common: names the marker file they meet through.  Nothing node-local goes
    into it when the launcher hands out an id, so that ranks on several nodes (torchrun --nnodes > 1, srun) sharing a
    file system compute the same name: HYDRA_PSPEC_RUN_ID as given; otherwise torchrun's rendezvous (MASTER_ADDR,
    MASTER_PORT, TORCHELASTIC_RUN_ID, restart count).  Those repeat from launch to launch: a marker left behind by a
    launch that was killed is told apart by its age (START_SKEW) and rank 0's nonce, see main().  Only a launch with
    neither (ranks started by hand from one shell) falls back to the parent's pid and start time."""
    env = os.environ
    if env.get("HYDRA_PSPEC_RUN_ID"):
        return f"run-{env['HYDRA_PSPEC_RUN_ID']}"
    if env.get("TORCHELASTIC_RUN_ID") or env.get("MASTER_PORT"):
        return "rdzv-" + "-".join(str(env.get(k, "")) for k in ("MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID",
                                                                 "TORCHELASTIC_RESTART_COUNT")).replace("/", "_")
    return f"pp{os.getppid()}-{_parent_start_ticks()}"


# how much earlier than this rank rank 0 may have started (launchers start ranks within seconds of each other; a marker
# older than that belongs to another launch)
START_SKEW = float(os.environ.get("HYDRA_PSPEC_START_SKEW", "120"))


def write_json_atomic(path, obj):
    tmp = Path(str(path) + f".tmp{os.getpid()}")
    with open(tmp, "w") as f:
        json.dump(obj, f, indent=2)
    os.replace(tmp, path)


def wait_for_file(path, newer_than, timeout=900.0, what="", nonce=None):
    """Poll for a JSON file written (atomically) after `newer_than` by another rank of this launch; with `nonce`,
    only a file that echoes this launch's nonce (handed out by rank 0 through the marker file) counts."""
    t_end = time.time() + timeout
    while time.time() < t_end:
        try:
            with open(path) as f:
                obj = json.load(f)
            if obj.get("created", 0.0) >= newer_than and (nonce is None or obj.get("nonce") == nonce):
                return obj
        except (FileNotFoundError, json.JSONDecodeError):
            pass
        time.sleep(0.05)
    raise SystemExit(f"rank synchronisation timed out waiting for {what or path}")


def _mem_available():
    """MemAvailable of /proc/meminfo in bytes (the host memory a run may still take)."""
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) * 1024
    except OSError:
        pass
    return 16e9


def load_aux(path, file_name, bl_str):
    """'file or directory' convention of the reference (:248-266): a directory means
    ``<dir>/<ant1>-<ant2>/<file_name>``."""
    fp = Path(path)
    if fp.is_dir():
        return np.load(fp / bl_str / file_name)
    return np.load(fp)


def main(argv=None):
    args = parse_args(argv)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    t_start = time.perf_counter()
    t_launch = time.time()

    from hydra_pspec_amd import pspec, synthetic, utils
    from hydra_pspec_amd.sharding import block_range

    # ---- inputs of this rank's block --------------------------------------------------
    freq_str = None          # known for UVH5 inputs only: names default files / directories as the reference does
    if args.synthetic:
        nbl_all, T, N = (int(v) for v in args.synthetic.split(","))
        lo, hi = block_range(nbl_all, world, rank)
        d = synthetic.make_baselines(N, T, args.Nfgmodes, k0=lo, nbl=hi - lo, dense=False)
        vis, flags_td = d["vis"], np.broadcast_to(~d["flags"][:, None, :], d["vis"].shape)
        antpairs = [(0, k + 1) for k in range(lo, hi)]
        default_fg, default_ps0, default_ninv = d["fgmodes"], d["ps0"], d["ninv_diag"]
    else:
        if not args.file_paths:
            raise SystemExit("Must pass file(s) to analyze via --file_paths.  Exiting.")
        fp = Path(args.file_paths[0])
        if fp.suffix in (".npy", ".npz"):
            cube = np.load(fp)
            if fp.suffix == ".npz":
                vis_all = cube["vis"]
                pairs_all = [tuple(p) for p in cube["antpairs"]] if "antpairs" in cube else None
                flags_all = cube["flags"] if "flags" in cube else None
            else:
                vis_all, pairs_all, flags_all = cube, None, None
            nbl_all = vis_all.shape[0]
            lo, hi = block_range(nbl_all, world, rank)
            vis = np.array(vis_all[lo:hi], dtype=complex)
            flags_td = np.zeros(vis.shape, bool) if flags_all is None else np.array(flags_all[lo:hi], bool)
            antpairs = pairs_all[lo:hi] if pairs_all else [(0, k + 1) for k in range(lo, hi)]
        else:
            from hydra_pspec_amd import uvh5
            assert len(args.file_paths) == 1, "one UVH5 file at a time"
            with uvh5.UVH5File(fp) as u:
                pairs_all = u.antpairs(args.ant_str)
                nbl_all = len(pairs_all)
                lo, hi = block_range(nbl_all, world, rank)
                fmask = uvh5.filter_freqs(args.freq_range, u.freqs_hz / 1e6) if args.freq_range else None
                antpairs = pairs_all[lo:hi]
                vis, flags_td = u.read_baselines(antpairs, fmask)
                fkeep = u.freqs_hz if fmask is None else u.freqs_hz[fmask]
                freq_str = f"{fkeep.min() / 1e6:.3f}-{fkeep.max() / 1e6:.3f}MHz"       # reference :332-335
        T, N = vis.shape[1:]
        default_fg = default_ps0 = default_ninv = None
    nbl = vis.shape[0]
    t_load = time.perf_counter() - t_start

    # ---- per-baseline auxiliary inputs (reference :379-460) ----------------------------
    ps0 = np.empty((nbl, N))
    S_general = None
    ninv = np.empty((nbl, N))
    ninv_dense = None        # (nbl,N,N) once any baseline has a non-diagonal inverse noise covariance
    fg = None
    flags_any = np.empty((nbl, N), bool)
    flags_pt = np.empty((nbl, T, N), bool) if args.per_time_flags else None
    for b, ap in enumerate(antpairs):
        bl = f"{ap[0]}-{ap[1]}"
        if args.flags:
            fl = load_aux(args.flags, args.flags_file, bl).astype(bool)
            assert fl.shape == vis[b].shape, "flags array does not match the per-baseline data"
        else:
            fl = flags_td[b]
        flags_any[b] = any_time_unflagged(~fl)
        if flags_pt is not None:
            flags_pt[b] = ~fl
        if args.noise:
            noise = load_aux(args.noise, args.noise_file, bl)
            if args.nsamples:
                noise = noise / np.sqrt(load_aux(args.nsamples, args.nsamples_file, bl))
            vis[b] = vis[b] + noise                                   # :415
        if args.sigcov0:
            S0 = load_aux(args.sigcov0, args.sigcov0_file, bl)
            p0, resid = pspec.pspec_from_covariance(S0)
            if resid > pspec.FOURIER_FORM_TOL:
                if S_general is None:
                    S_general = np.zeros((nbl, N, N), dtype=complex)
                S_general[b] = S0
            ps0[b] = p0
        else:
            ps0[b] = default_ps0 if default_ps0 is not None else float(N)   # sigcov0 = eye(N) (:425)
        if args.noise_cov:
            # the reference hands the FULL inverse to the sampler (run-hydra-pspec.py:436); a matrix
            # with off-diagonal terms is never reduced to its diagonal here
            Ni_b = np.linalg.inv(load_aux(args.noise_cov, args.noise_cov_file, bl))
            ninv[b] = np.diag(Ni_b).real
            if np.any(Ni_b - np.diag(np.diag(Ni_b)) != 0):
                if ninv_dense is None:
                    ninv_dense = np.zeros((nbl, N, N), dtype=complex)
                    for bb in range(b):
                        ninv_dense[bb] = np.diag(ninv[bb])
                ninv_dense[b] = Ni_b
            elif ninv_dense is not None:
                ninv_dense[b] = np.diag(ninv[b])
        else:
            ninv[b] = default_ninv[b] if default_ninv is not None else 1.0 / 10.0 ** 2   # :438
            if ninv_dense is not None:
                ninv_dense[b] = np.diag(ninv[b])
        if args.fgmodes:
            # default file name of scripts/calc-vis-cov-matrices.py (reference :444-449)
            default_name = f"evecs-{freq_str}.npy" if freq_str else "fgmodes.npy"
            f_b = load_aux(args.fgmodes, args.fgmodes_file or default_name, bl)[:, :args.Nfgmodes]
        elif default_fg is not None:
            f_b = default_fg
        else:   # Legendre polynomials (:456-460)
            import scipy.special
            f_b = np.array([scipy.special.legendre(i)(np.linspace(-1., 1., N))
                            for i in range(args.Nfgmodes)]).T
        if fg is None:
            fg = np.empty((nbl,) + f_b.shape, dtype=complex)
        fg[b] = f_b
    ps_prior = make_ps_prior(N, args.n_ps_prior_bins, args.ps_prior_lo, args.ps_prior_hi)

    # ---- output tree ---------------------------------------------------------------------
    out_dir = Path(args.out_dir)
    if args.dirname:                                       # reference :336-342
        dirname = args.dirname + ("-map-estimate" if args.map_estimate else "")
    elif freq_str:
        dirname = f"results-{freq_str}-Niter-{args.Niter}"
    else:
        dirname = f"results-seed-{args.seed}-Niter-{args.Niter}"
    if args.dry_run:
        # a dry run writes all-zero chains: it gets a tree of its own, so that neither --clobber nor --resume can
        # ever mix them with real results (and `dry_run` is one of the arguments a --resume must match)
        dirname = "dryrun-" + dirname
    results = out_dir / dirname
    # Rank 0 alone prepares the tree (moves an earlier one aside, creates the directory) and reads the earlier
    # run's args.json for --resume; the other ranks wait for its marker file before they touch the tree and take
    # the earlier arguments from it (no rank reads args.json while rank 0 rewrites it).  The reference does this
    # with MPI barriers / bcast (run-hydra-pspec.py:343-366); here the ranks of a launch meet through files.
    marker = out_dir / f".{dirname}.{launch_token()}.ranks.json"
    old_args = None
    nonce = None
    if rank == 0:
        # leftovers of a launch that died (same token: torchrun's default run id and port repeat): gone before
        # anything of this launch can be mistaken for them
        import uuid
        nonce = uuid.uuid4().hex
        for stale in [marker] + (sorted(results.glob(".timings-*.json")) if results.exists() else []):
            try:
                os.remove(stale)
            except FileNotFoundError:
                pass
        if args.resume and (results / "args.json").exists():
            with open(results / "args.json") as f:
                old_args = json.load(f)
        if results.exists() and not args.clobber and not args.resume:
            # keep earlier results: move them aside under their modification time (reference :343-345,
            # utils.add_mtime_to_filepath)
            from datetime import datetime
            import shutil
            mtime = datetime.fromtimestamp(os.path.getmtime(results)).isoformat()
            shutil.move(str(results), str(results.with_name(f"{results.name}-{mtime}")))
        results.mkdir(parents=True, exist_ok=True)
        if world > 1:
            write_json_atomic(marker, {"created": time.time(), "old_args": old_args, "nonce": nonce})
    else:
        # only a marker written after this launch began counts (rank 0 removes the marker of a finished or failed launch
        # on its way out, see below; one left by a KILLED launch is older than this rank's start minus the skew unless
        # the relaunch came within START_SKEW seconds -- give such relaunches a fresh HYDRA_PSPEC_RUN_ID)
        met = wait_for_file(marker, t_launch - START_SKEW, what="rank 0 to prepare the output tree")
        old_args, nonce = met["old_args"], met.get("nonce")

    # ---- sampling -----------------------------------------------------------------------
    try:
        return sample_and_write(args, rank, world, local_rank, results, marker, nonce, old_args, locals())
    except BaseException as e:      # SystemExit, and what the library raises (FloatingPointError, RuntimeError, ...)
        if rank > 0:        # rank 0 waits for this rank's timings file: tell it not to
            try:
                write_json_atomic(results / f".timings-{rank}.json",
                                  {"created": time.time(), "nonce": nonce, "rank": rank, "failed": repr(e)})
            except OSError:
                pass
        raise
    finally:
        if rank == 0 and world > 1:     # also when this launch failed: no marker outlives its launch
            try:
                os.remove(marker)
            except FileNotFoundError:
                pass


def sample_and_write(args, rank, world, local_rank, results, marker, nonce, old_args, env):
    """Everything after the output tree exists: the chains, the periodic writes, the merged timings."""
    import shutil
    from hydra_pspec_amd import drain, pspec
    (vis, flags_any, flags_pt, fg, ninv, ninv_dense, ps_prior, ps0, S_general, antpairs, nbl, nbl_all, N, T, t_load,
     t_start, t_launch) = (env[k] for k in ("vis", "flags_any", "flags_pt", "fg", "ninv", "ninv_dense", "ps_prior",
                                            "ps0", "S_general", "antpairs", "nbl", "nbl_all", "N", "T", "t_load",
                                            "t_start", "t_launch"))
    import torch
    if not args.dry_run:
        torch.cuda.set_device(local_rank)
    all_out = args.outputs == "all"
    keep = ("signal_cr", "fg_amps", "chisq") if all_out else ()
    t0 = time.perf_counter()
    if S_general is not None and np.any(np.abs(S_general).sum(axis=(1, 2)) == 0):
        raise SystemExit("mixing Fourier-form and general sigcov0 across baselines is not supported")
    Niter = 1 if args.map_estimate else int(args.Niter)
    chunk = max(1, int(args.write_Niter))
    thin = max(1, int(args.thin))
    if thin > 1 and chunk % thin:
        raise SystemExit(f"--write_Niter {chunk} must be a multiple of --thin {thin}")
    names = ["signal_ps", "ln_post"] + (["signal_cr", "fg_amps", "chisq"] if all_out else [])
    bdirs = [results / f"{ap[0]}-{ap[1]}" for ap in antpairs]
    M = int(fg.shape[-1])

    # ---- what the run will write and stage, BEFORE anything is sampled (SURVEY 7.3 "output volume": at C3, --outputs
    # all without thinning is 537 GB of gcr-eor.npy) ------------------------------------------------------------------
    disk_b, stage_b = drain.estimate_bytes(names, nbl, T, N, M, Niter, min(chunk, Niter), thin)
    budget = args.host_mem_gb * 1e9 if args.host_mem_gb else 0.5 * _mem_available()
    try:
        free_disk = shutil.disk_usage(results).free
    except OSError:
        free_disk = None
    if rank == 0 and (args.verbose or 2 * stage_b > 0.25 * budget):
        print(f"outputs of this rank: {disk_b / 1e9:.2f} GB on disk ({len(names)} histories x {nbl} baselines x {Niter} "
              f"iterations, thin {thin}); staged per flush: {stage_b / 1e9:.2f} GB on the device and twice that in pinned "
              f"host memory (budget {budget / 1e9:.1f} GB)", flush=True)
    hint = ("choose --outputs ps, a larger --thin or a smaller --write_Niter" if all_out
            else "choose a smaller --write_Niter")
    if 2 * stage_b > budget:
        raise SystemExit(f"the output drain would need 2 x {stage_b / 1e9:.2f} GB of pinned host memory per flush "
                         f"(--write_Niter {chunk}, --thin {thin}, --outputs {args.outputs}) against a budget of "
                         f"{budget / 1e9:.1f} GB (--host_mem_gb): {hint}")
    if free_disk is not None and disk_b > free_disk:
        raise SystemExit(f"the run would write {disk_b / 1e9:.2f} GB into {results} which has {free_disk / 1e9:.2f} GB "
                         f"free: {hint}")

    # --resume: continue from the checkpoints in the output tree.  They must come from the same
    # inputs and seed (args.json); baselines caught mid-write at different iterations are rolled
    # back to the earliest one (the chain is deterministic, nothing is lost but time).
    iter0 = 0
    if args.resume and (args.map_estimate or S_general is not None):
        raise SystemExit("--resume is not available with --map_estimate or a sigcov0 that is not of the form "
                         "Fop^H diag(p) Fop: such a run cannot be continued from its bandpowers")
    if args.resume:
        if old_args is None:
            raise SystemExit(f"--resume: no args.json of an earlier run in {results}")
        run_only = {"Niter", "resume", "clobber", "verbose", "write_Niter", "Nproc", "config", "outputs", "host_mem_gb",
                    "fsync", "write_workers"}
        diff = sorted(k for k in vars(args) if k not in run_only and old_args.get(k, build_parser().get_default(k))
                      != getattr(args, k))
        if diff:
            raise SystemExit("--resume: the earlier run in " + str(results) + " used different "
                             + ", ".join(f"{k} ({old_args.get(k)!r} != {getattr(args, k)!r})" for k in diff)
                             + ": refusing to splice two different chains")
        k_done = None
        for bdir in bdirs:
            missing = [drain.HISTORY_FILES[k] for k in names if not (bdir / drain.HISTORY_FILES[k]).exists()]
            if missing:
                raise SystemExit(f"--resume: {bdir} has no {', '.join(missing)} (an earlier run with --outputs ps "
                                 "cannot be continued with --outputs all)")
            kb = drain.checkpoint_rows(bdir, names, thin)
            k_done = kb if k_done is None else min(k_done, kb)
        k_done = k_done or 0
        if k_done >= Niter:
            raise SystemExit(f"--resume: the chains in {results} already hold {k_done} iterations (--Niter {Niter}): "
                             "nothing to do")
        if k_done > 0:
            iter0 = k_done
            ps0 = np.stack([np.load(bdir / "dps-eor.npy", mmap_mode="r")[k_done - 1] for bdir in bdirs])
        elif rank == 0:
            print("--resume: the checkpoints hold no iteration yet; starting at 0", flush=True)
    if rank == 0:
        write_json_atomic(results / "args.json", vars(args))

    Ninv_arg = ninv if ninv_dense is None else ninv_dense
    gb = None
    if not args.dry_run:
        # Several ranks: never the split factorisation.  Its order of operations depends on the size of the batch, so a
        # baseline's chain would depend (in the last bits) on how many ranks the baselines were dealt to -- a --resume
        # with another Nproc must continue the very same chain -- and ranks that share a GPU (rehearsals, more ranks
        # than devices) could starve each other's co-operating workgroups (hpx.h, HPX_OPT_FACTOR_SPLIT).
        gb = pspec.make_batch(vis, flags_any if flags_pt is None else flags_pt, fg, Ninv_arg, ps_prior, Niter,
                              seed=args.seed, map_estimate=args.map_estimate, allow_split=(world == 1), solver=args.solver)
        free_dev = torch.cuda.mem_get_info()[0]
        if 2 * stage_b > free_dev:          # the chunk being sampled + the one being copied
            gb.close()
            raise SystemExit(f"two chunks of outputs ({2 * stage_b / 1e9:.2f} GB) do not fit the {free_dev / 1e9:.1f} GB "
                             f"left on the device: {hint}")
    out_drain = drain.ChainDrain(torch, bdirs, names, T, N, M, min(chunk, Niter), thin=thin, all_out=all_out,
                                 use_gpu=gb is not None, fsync=args.fsync, workers=args.write_workers)
    out_drain.start(iter0)
    t_setup = time.perf_counter() - t0

    done = iter0
    if gb is not None:
        gb.iter_done = iter0
    t_process = 0.0
    shapes = {"signal_ps": (N,), "ln_post": (), "signal_cr": (T, N), "fg_amps": (T, M), "chisq": (T, N)}
    failed = True
    try:
        while done < Niter:
            n = min(chunk - done % chunk, Niter - done)
            tp = time.perf_counter()
            if gb is None:            # --dry_run: files of the right shapes, no sampling
                out = {k: torch.zeros((nbl, -(-n // thin) if k in drain.THINNED else n) + shapes[k],
                                      dtype=torch.complex128 if k in ("signal_cr", "fg_amps") else torch.float64)
                       for k in names}
            elif done == 0 and S_general is not None:
                shp0 = np.stack([pspec.sqrt_cov_delay_basis(S_general[b]) for b in range(nbl)])
                out = gb.run(n, shp0=shp0, keep=keep, thin=thin)
            else:
                out = gb.run(n, ps0=ps0 if done == iter0 else None, keep=keep, thin=thin)
            done += n
            # the chunk is complete on the device: its copy and its files proceed behind the next chunk's sampling
            out_drain.submit(out, n, done, periodic=(done % chunk == 0))
            del out
            t_process += time.perf_counter() - tp
            if args.verbose and rank == 0:
                print(f"iteration {done}/{Niter}: {nbl * n / (time.perf_counter() - tp):.1f} baseline*iter/s", flush=True)
        failed = False
    finally:
        t_tail = time.perf_counter()
        try:
            out_drain.close(abort=failed)       # (complete chunks still reach the disk when the sampler failed)
        finally:
            if gb is not None:
                gb.close()
        t_tail = time.perf_counter() - t_tail
    write_times, ant_strs = out_drain.write_times, [f"{ap[0]}_{ap[1]}" for ap in antpairs]

    # every rank's write times reach rank 0 (the reference gathers them, run-hydra-pspec.py:557, and writes
    # one timings.json, :570-581): per-rank files, merged and removed by rank 0
    mine = {"created": time.time(), "nonce": nonce, "rank": rank, "ant_pairs": ant_strs, "write_times": write_times}
    if rank > 0:
        write_json_atomic(results / f".timings-{rank}.json", mine)
    if rank == 0:
        t_bar = time.perf_counter()
        write_data = [{k: mine[k] for k in ("rank", "ant_pairs", "write_times")}]
        for r in range(1, world):
            other = wait_for_file(results / f".timings-{r}.json", t_launch - START_SKEW, what=f"rank {r} to finish",
                                  nonce=nonce)
            if "failed" in other:
                raise SystemExit(f"rank {r} stopped: {other['failed']}")
            write_data.append({k: other[k] for k in ("rank", "ant_pairs", "write_times")})
            os.remove(results / f".timings-{r}.json")
        t_bar = time.perf_counter() - t_bar
        total = time.perf_counter() - t_start
        timings = {"num_ranks": world, "num_baselines": int(nbl_all),
                   "rank_0_timers": {"load_data": t_load, "scatter": 0.0, "process": t_process,
                                     "barrier": t_bar, "total": total},
                   "write_data": write_data}
        write_json_atomic(results / "timings.json", timings)
        # beyond the reference's keys (kept out of timings.json, whose key set is the reference's): where the rest of
        # `total` went and what the drain did
        write_json_atomic(results / "drain.json", {
            "setup_s": t_setup, "drain_tail_s": t_tail, "bytes_written": int(out_drain.bytes_written),
            "writer_file_s": out_drain.t_files, "writer_copy_wait_s": out_drain.t_copy_wait,
            "sampler_backpressure_s": out_drain.t_backpressure, "write_workers": out_drain.workers,
            "fsync": bool(args.fsync), "thin": thin, "write_Niter": chunk, "iter0": iter0, "Niter": Niter,
            "baselines_this_rank": int(nbl),
            "baseline_iter_per_s_process": (nbl * (Niter - iter0) / t_process) if t_process > 0 else None})
        ru = getrusage(RUSAGE_SELF)
        with open(results / "resources.json", "w") as f:
            json.dump({"ru_maxrss": ru.ru_maxrss, "ru_utime": ru.ru_utime, "ru_stime": ru.ru_stime}, f, indent=2)
        if args.verbose:
            print(f"{nbl_all} baselines x {args.Niter} iterations: process {t_process:.2f} s "
                  f"({nbl * args.Niter / t_process:.1f} baseline*iter/s on this rank)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
