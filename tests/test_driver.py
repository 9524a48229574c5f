"""run-hydra-pspec.py counterpart: argument surface / host logic on CPU, end-to-end on the GPU."""
import importlib.util
import json
from pathlib import Path

import numpy as np
import pytest
import yaml

from conftest import REPO


def _driver():
    spec = importlib.util.spec_from_file_location("run_hydra_pspec", REPO / "run-hydra-pspec.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_flag_names_and_defaults_match_reference():
    drv = _driver()
    a = drv.parse_args(["--file_paths", "x.npy"])
    # reference defaults: run-hydra-pspec.py:40-234
    assert (a.ant_str, a.Nfgmodes, a.n_ps_prior_bins, a.ps_prior_lo, a.ps_prior_hi) == ("cross", 8, 3, 0.0, 0.0)
    assert (a.Niter, a.seed, a.Nproc, a.out_dir, a.write_Niter, a.clobber, a.map_estimate) == \
        (100, None, 1, "./", 100, False, False)
    for k in ("sigcov0", "sigcov0_file", "fgmodes", "fgmodes_file", "freq_range", "flags", "flags_file", "noise",
              "noise_file", "noise_cov", "noise_cov_file", "nsamples", "nsamples_file", "dirname", "verbose"):
        assert hasattr(a, k)


def test_config_file_keys(tmp_path):
    drv = _driver()
    cfg = dict(ant_str="0_1", seed=7123689, Niter=1000, Nproc=2, verbose=True, ps_prior_lo=0.1, ps_prior_hi=2,
               n_ps_prior_bins=3, dirname="results-x", clobber=False, sigcov0="./eor-cov.npy", Nfgmodes=12,
               fgmodes="./fgmodes.npy", noise="./noise.npy", noise_cov="./noise-cov.npy",
               file_paths=["./vis.npy"])          # the keys of test_data/config.yaml
    (tmp_path / "c.yaml").write_text(yaml.safe_dump(cfg))
    a = drv.parse_args(["--config", str(tmp_path / "c.yaml"), "--Niter", "200"])
    assert a.Niter == 200 and a.seed == 7123689 and a.Nfgmodes == 12 and a.ps_prior_hi == 2
    assert a.file_paths == ["./vis.npy"]
    (tmp_path / "bad.yaml").write_text("not_a_key: 1\n")
    with pytest.raises(SystemExit):
        drv.parse_args(["--config", str(tmp_path / "bad.yaml")])


def test_prior_and_flag_reduction():
    drv = _driver()
    pr = drv.make_ps_prior(120, 3, 0.1, 2.0)
    assert pr[0, 57:64].tolist() == [2.0] * 7 and pr[1, 57:64].tolist() == [0.1] * 7 and pr[:, :57].sum() == 0
    assert drv.make_ps_prior(120, 3, 0.0, 0.0).sum() == 0
    w = np.ones((5, 6), bool)
    w[2, 4] = False
    assert drv.any_time_unflagged(w).tolist() == [True, True, True, True, False, True]


@pytest.mark.gpu
def test_driver_end_to_end_config1(golden, tmp_path):
    """test_data-like directory layout through the driver: same chain as the reference."""
    drv = _driver()
    g = golden("chain_testdata")
    aux = tmp_path / "aux" / "0-1"
    aux.mkdir(parents=True)
    np.save(aux / "eor-cov.npy", g["S_initial"])
    np.save(aux / "fgmodes.npy", np.pad(g["fgmodes"], ((0, 0), (0, 3))))      # extra modes are cut (:453)
    np.save(aux / "noise-cov.npy", np.diag(1.0 / g["ninv_diag"]))
    np.savez(tmp_path / "vis.npz", vis=g["vis"][None], antpairs=np.array([[0, 1]]))
    rc = drv.main(["--file_paths", str(tmp_path / "vis.npz"), "--sigcov0", str(tmp_path / "aux"),
                   "--sigcov0_file", "eor-cov.npy", "--fgmodes", str(tmp_path / "aux"), "--fgmodes_file",
                   "fgmodes.npy", "--Nfgmodes", "12", "--noise_cov", str(tmp_path / "aux"), "--noise_cov_file",
                   "noise-cov.npy", "--ps_prior_lo", "0.1", "--ps_prior_hi", "2", "--seed", "7123689",
                   "--Niter", "20", "--out_dir", str(tmp_path), "--dirname", "res"])
    assert rc == 0
    bdir = tmp_path / "res" / "0-1"
    ps = np.load(bdir / "dps-eor.npy")
    assert ps.shape == (20, 120)
    assert np.max(np.abs(ps / g["ref_ps"][:20] - 1)) < 1e-6
    assert np.load(bdir / "gcr-eor.npy").shape == (20, 203, 120)
    assert np.load(bdir / "fg-amps.npy").shape == (20, 203, 12)
    assert np.load(bdir / "chisq.npy").shape == (20, 203, 120)
    assert np.load(bdir / "ln-post.npy").shape == (20,)
    assert np.load(bdir / "cov-eor.npy").shape == (120, 120)
    t = json.loads((tmp_path / "res" / "timings.json").read_text())
    assert set(t) == {"num_ranks", "num_baselines", "rank_0_timers", "write_data"}
    assert set(t["rank_0_timers"]) == {"load_data", "scatter", "process", "barrier", "total"}
    assert (tmp_path / "res" / "args.json").exists()


@pytest.mark.gpu
def test_driver_synthetic_batch(tmp_path):
    drv = _driver()
    rc = drv.main(["--synthetic", "3,8,32", "--Nfgmodes", "4", "--ps_prior_lo", "0.1", "--ps_prior_hi", "2",
                   "--seed", "5", "--Niter", "4", "--out_dir", str(tmp_path), "--dirname", "r", "--outputs", "ps"])
    assert rc == 0
    for k in (1, 2, 3):
        assert np.load(tmp_path / "r" / f"0-{k}" / "dps-eor.npy").shape == (4, 32)
        assert not (tmp_path / "r" / f"0-{k}" / "gcr-eor.npy").exists()


@pytest.mark.gpu
def test_driver_resume(tmp_path):
    drv = _driver()
    common = ["--synthetic", "2,8,32", "--Nfgmodes", "4", "--ps_prior_lo", "0.1", "--ps_prior_hi", "2", "--seed", "5",
              "--out_dir", str(tmp_path)]
    assert drv.main(common + ["--Niter", "6", "--dirname", "full"]) == 0
    assert drv.main(common + ["--Niter", "3", "--dirname", "part"]) == 0
    assert drv.main(common + ["--Niter", "6", "--dirname", "part", "--resume"]) == 0
    for k in (1, 2):
        for f in ("dps-eor.npy", "gcr-eor.npy", "ln-post.npy", "chisq.npy", "fg-amps.npy"):
            assert np.array_equal(np.load(tmp_path / "part" / f"0-{k}" / f), np.load(tmp_path / "full" / f"0-{k}" / f))


@pytest.mark.gpu
def test_driver_config1_from_uvh5(golden, tmp_path):
    """Config 1 straight from the reference's UVH5 test file (tests/golden/vis-eor-fgs.uvh5, a copy
    of test_data/vis-eor-fgs.uvh5) through the package's own HDF5 reader: XX+YY of baseline (0, 1)
    plus the noise realisation reproduces the reference chain (run-hydra-pspec.py:305-322, :415)."""
    from hydra_pspec_amd import uvh5
    drv = _driver()
    g = golden("chain_testdata")
    src = Path(__file__).parent / "golden" / "vis-eor-fgs.uvh5"
    pairs, vis, flags, ntot, _ = uvh5.read_uvh5_block(src, 0, 1, ant_str="cross")
    assert pairs == [(0, 1)] and ntot == 1 and not flags.any()
    aux = tmp_path / "aux" / "0-1"
    aux.mkdir(parents=True)
    np.save(aux / "noise.npy", g["vis"] - vis[0])               # the golden cube is XX+YY + noise.npy
    np.save(aux / "eor-cov.npy", g["S_initial"])
    np.save(aux / "fgmodes.npy", g["fgmodes"])
    np.save(aux / "noise-cov.npy", np.diag(1.0 / g["ninv_diag"]))
    rc = drv.main(["--file_paths", str(src), "--sigcov0", str(tmp_path / "aux"), "--sigcov0_file", "eor-cov.npy",
                   "--fgmodes", str(tmp_path / "aux"), "--fgmodes_file", "fgmodes.npy", "--Nfgmodes", "12",
                   "--noise", str(tmp_path / "aux"), "--noise_file", "noise.npy",
                   "--noise_cov", str(tmp_path / "aux"), "--noise_cov_file", "noise-cov.npy",
                   "--ps_prior_lo", "0.1", "--ps_prior_hi", "2", "--seed", "7123689", "--Niter", "10",
                   "--out_dir", str(tmp_path), "--dirname", "res"])
    assert rc == 0
    ps = np.load(tmp_path / "res" / "0-1" / "dps-eor.npy")
    assert np.max(np.abs(ps / g["ref_ps"][:10] - 1)) < 1e-6


@pytest.mark.gpu
def test_driver_output_naming_and_no_clobber(tmp_path):
    """Directory naming of the reference driver (:336-345): '<dirname>-map-estimate', the
    'results-<fmin>-<fmax>MHz-Niter-<n>' default for UVH5 input, and earlier results moved aside
    under their mtime unless --clobber."""
    drv = _driver()
    src = Path(__file__).parent / "golden" / "mini.uvh5"
    common = ["--file_paths", str(src), "--Nfgmodes", "3", "--Niter", "2", "--seed", "3", "--out_dir", str(tmp_path),
              "--ant_str", "cross", "--outputs", "ps"]
    assert drv.main(common) == 0
    res = tmp_path / "results-100.000-105.500MHz-Niter-2"
    assert (res / "0-1" / "dps-eor.npy").exists() and (res / "1-2" / "dps-eor.npy").exists()
    assert drv.main(common) == 0                                   # again: the first tree is kept aside
    moved = [p for p in tmp_path.iterdir() if p.name.startswith(res.name + "-")]
    assert len(moved) == 1 and (moved[0] / "0-1" / "dps-eor.npy").exists() and (res / "0-2" / "dps-eor.npy").exists()
    assert drv.main(common + ["--clobber"]) == 0
    assert len([p for p in tmp_path.iterdir() if p.name.startswith(res.name)]) == 2
    assert drv.main(common + ["--dirname", "run", "--map_estimate"]) == 0
    assert (tmp_path / "run-map-estimate" / "0-1" / "dps-eor.npy").exists()


@pytest.mark.gpu
def test_driver_periodic_checkpoints_and_resume_guard(tmp_path, monkeypatch):
    """--write_Niter checkpoints are on disk while the run is still going (reference pspec.py:625-636),
    an interrupted run continues from them to the chain of an uninterrupted one, and --resume refuses
    an output tree written with another seed."""
    from hydra_pspec_amd import pspec
    drv = _driver()
    common = ["--synthetic", "2,8,32", "--Nfgmodes", "4", "--ps_prior_lo", "0.1", "--ps_prior_hi", "2",
              "--out_dir", str(tmp_path), "--write_Niter", "2"]
    assert drv.main(common + ["--seed", "5", "--Niter", "6", "--dirname", "full"]) == 0
    # crash after the second chunk: the third gb.run raises
    real_run, calls = pspec.GibbsBatch.run, []

    seen_mid_run = []

    def dying_run(self, niter, **kw):
        calls.append(niter)
        if len(calls) == 3:
            # MID-RUN: the sampler is still "busy" (here: polling) while the drain's writer thread brings the first two
            # chunks to disk behind it; whatever np.load sees at any instant is a valid array of complete iterations
            import time
            t_end = time.time() + 20
            while time.time() < t_end:
                a = np.load(tmp_path / "part" / "0-1" / "dps-eor.npy")
                assert a.ndim == 2 and a.shape[1] == 32 and a.shape[0] in (0, 2, 4)
                seen_mid_run.append(a.shape[0])
                if a.shape[0] == 4:
                    break
                time.sleep(0.01)
            raise RuntimeError("simulated crash")
        return real_run(self, niter, **kw)
    monkeypatch.setattr(pspec.GibbsBatch, "run", dying_run)
    with pytest.raises(RuntimeError, match="simulated crash"):
        drv.main(common + ["--seed", "5", "--Niter", "6", "--dirname", "part"])
    monkeypatch.setattr(pspec.GibbsBatch, "run", real_run)
    assert calls == [2, 2, 2] and seen_mid_run[-1] == 4
    full_ps = np.load(tmp_path / "full" / "0-1" / "dps-eor.npy")
    assert np.array_equal(np.load(tmp_path / "part" / "0-1" / "dps-eor.npy"), full_ps[:4])
    for k in (1, 2):
        assert np.load(tmp_path / "part" / f"0-{k}" / "dps-eor.npy").shape == (4, 32)
        assert np.load(tmp_path / "part" / f"0-{k}" / "cov-eor.npy").shape == (4, 32)     # rows [:done], periodic write
    with pytest.raises(SystemExit, match="seed"):
        drv.main(common + ["--seed", "6", "--Niter", "6", "--dirname", "part", "--resume"])
    assert drv.main(common + ["--seed", "5", "--Niter", "6", "--dirname", "part", "--resume"]) == 0
    for k in (1, 2):
        for f in ("dps-eor.npy", "gcr-eor.npy", "ln-post.npy", "chisq.npy", "fg-amps.npy", "cov-eor.npy"):
            assert np.array_equal(np.load(tmp_path / "part" / f"0-{k}" / f), np.load(tmp_path / "full" / f"0-{k}" / f)), f


@pytest.mark.gpu
def test_driver_correlated_noise_cov(golden, tmp_path):
    """--noise_cov with off-diagonal terms reaches the sampler as the full inverse (reference
    run-hydra-pspec.py:436), never as its diagonal (run-hydra-pspec.py:199)."""
    drv = _driver()
    g = golden("steps_dense")
    aux = tmp_path / "aux" / "0-1"
    aux.mkdir(parents=True)
    np.save(aux / "eor-cov.npy", g["in_S"])
    np.save(aux / "fgmodes.npy", g["in_fgmodes"])
    np.save(aux / "noise-cov.npy", g["in_noise_cov"])
    np.savez(tmp_path / "vis.npz", vis=g["in_vis"][None], antpairs=np.array([[0, 1]]))
    rc = drv.main(["--file_paths", str(tmp_path / "vis.npz"), "--sigcov0", str(tmp_path / "aux"), "--sigcov0_file",
                   "eor-cov.npy", "--fgmodes", str(tmp_path / "aux"), "--fgmodes_file", "fgmodes.npy", "--Nfgmodes", "4",
                   "--noise_cov", str(tmp_path / "aux"), "--noise_cov_file", "noise-cov.npy", "--ps_prior_lo", "0.1",
                   "--ps_prior_hi", "2", "--seed", "77", "--Niter", "6", "--out_dir", str(tmp_path), "--dirname", "res"])
    assert rc == 0
    ps = np.load(tmp_path / "res" / "0-1" / "dps-eor.npy")
    assert np.max(np.abs(ps / g["chain_ps"] - 1)) < 1e-6
    # the same covariance with channels flagged in the input cube (all times, as the reference's any-time mask
    # reduces them, run-hydra-pspec.py:531-535): the reference's column-masked, non-Hermitian system -- its own
    # chain (golden fl_chain_ps) comes out of the driver
    fl = g["fl_flags"]
    flags_td = np.broadcast_to(~fl, g["in_vis"].shape)[None]                 # True = flagged sample
    np.savez(tmp_path / "visf.npz", vis=g["in_vis"][None], antpairs=np.array([[0, 1]]), flags=flags_td)
    rc = drv.main(["--file_paths", str(tmp_path / "visf.npz"), "--sigcov0", str(tmp_path / "aux"), "--sigcov0_file",
                   "eor-cov.npy", "--fgmodes", str(tmp_path / "aux"), "--fgmodes_file", "fgmodes.npy", "--Nfgmodes", "4",
                   "--noise_cov", str(tmp_path / "aux"), "--noise_cov_file", "noise-cov.npy", "--ps_prior_lo", "0.1",
                   "--ps_prior_hi", "2", "--seed", "77", "--Niter", "6", "--out_dir", str(tmp_path), "--dirname", "resf"])
    assert rc == 0
    ps = np.load(tmp_path / "resf" / "0-1" / "dps-eor.npy")
    assert np.max(np.abs(ps / g["fl_chain_ps"] - 1)) < 1e-6


@pytest.mark.gpu
def test_driver_per_time_flags(tmp_path):
    """--per_time_flags keeps (Nbl, Ntimes, Nfreqs) flags from the input cube time dependent
    instead of the reference driver's any-time reduction (run-hydra-pspec.py:524-541)."""
    from hydra_pspec_amd import pspec, synthetic
    drv = _driver()
    nbl, T, N, M = 2, 8, 32, 4
    d = synthetic.make_baselines(N, T, M, k0=3, nbl=nbl, dense=False)
    rng = np.random.default_rng(4)
    flagged = rng.uniform(size=(nbl, T, N)) < 0.1                        # True = flagged sample, as in UVH5
    fg = np.broadcast_to(d["fgmodes"], (nbl, N, M)).copy()
    for b in range(nbl):
        (tmp_path / "fg" / f"0-{b + 1}").mkdir(parents=True)
        np.save(tmp_path / "fg" / f"0-{b + 1}" / "fgmodes.npy", fg[b])
    np.savez(tmp_path / "vis.npz", vis=d["vis"], flags=flagged, antpairs=np.array([[0, 1], [0, 2]]))
    common = ["--file_paths", str(tmp_path / "vis.npz"), "--fgmodes", str(tmp_path / "fg"), "--Nfgmodes", str(M),
              "--seed", "5", "--Niter", "3", "--out_dir", str(tmp_path), "--outputs", "ps"]
    assert drv.main(common + ["--dirname", "pt", "--per_time_flags"]) == 0
    assert drv.main(common + ["--dirname", "any"]) == 0
    ninv = np.full((nbl, N), 1.0 / 100.0)
    want = pspec.gibbs_sample_with_fg_batched(d["vis"], ~flagged, fg, ninv, np.zeros((2, N)),
                                              ps_initial=np.full((nbl, N), float(N)), Niter=3, seed=5)
    for b in range(nbl):
        got = np.load(tmp_path / "pt" / f"0-{b + 1}" / "dps-eor.npy")
        assert np.array_equal(got, want["signal_ps"][b])
        assert not np.array_equal(got, np.load(tmp_path / "any" / f"0-{b + 1}" / "dps-eor.npy"))


def test_npy_appender_keeps_a_valid_file_at_every_instant(tmp_path):
    """hydra_pspec_amd/npy_append.py: rows appended flush by flush, the header's shape rewritten in place AFTER the data;
    numpy.load reads the complete prefix whatever state the tail is in; a resume cuts back to a row count and goes on;
    a file written by numpy.save is continued in place (numpy pads its headers) -- never a rewrite of the history."""
    import os
    from hydra_pspec_amd.npy_append import NpyAppender, read_header
    rng = np.random.default_rng(0)
    x = rng.standard_normal((7, 3, 4)) + 1j * rng.standard_normal((7, 3, 4))
    fn = tmp_path / "a.npy"
    f = NpyAppender(fn, (3, 4), np.complex128).start()
    assert np.load(fn).shape == (0, 3, 4)
    ino = os.stat(fn).st_ino
    f.append(x[:2])
    assert np.array_equal(np.load(fn), x[:2])
    # a torn flush: data beyond the header's row count (the next flush's rows, partially written) are ignored
    with open(fn, "ab") as raw:
        raw.write(b"\x01" * 100)
    assert np.array_equal(np.load(fn), x[:2])
    f.append(x[2:5])
    assert np.array_equal(np.load(fn), x[:5]) and os.stat(fn).st_ino == ino          # same file: appended, not replaced
    g = NpyAppender(fn, (3, 4), np.complex128).start(keep_rows=3)                     # resume at 3 iterations
    assert np.array_equal(np.load(fn), x[:3]) and os.path.getsize(fn) == read_header(fn)[2] + 3 * 12 * 16
    g.append(x[3:])
    assert np.array_equal(np.load(fn), x)
    with pytest.raises(ValueError):
        g.append(x[:, :2])
    np.save(tmp_path / "b.npy", x[:, 0, 0].real.copy())                               # numpy's own writer
    h = NpyAppender(tmp_path / "b.npy", (), np.float64).start(keep_rows=4)
    h.append(np.array([7.0, 8.0]))
    assert np.array_equal(np.load(tmp_path / "b.npy"), np.r_[x[:4, 0, 0].real, 7.0, 8.0])
    with pytest.raises(ValueError):
        NpyAppender(tmp_path / "b.npy", (), np.float64).start(keep_rows=9)


def test_circulant_cov_is_the_fourier_form():
    """drain.circulant_cov = Fop^H diag(p) Fop (reference pspec.py:313-322) for even and odd channel counts."""
    from hydra_pspec_amd import drain, utils
    for N in (8, 9, 120):
        p = np.random.default_rng(N).uniform(0.1, 2.0, N)
        F = utils.fourier_operator(N)
        assert np.abs(drain.circulant_cov(p) - F.conj().T @ np.diag(p) @ F).max() < 1e-11 * N


def test_driver_drain_thin_budget_and_append_only(tmp_path, capsys):
    """The driver's output side without a GPU (--dry_run): --thin K keeps every K-th iteration of the three large
    histories and every iteration of dps-eor / ln-post; history files are appended to (same inode across flushes and
    across a --resume), never rewritten; a run whose staged chunks exceed the host budget stops BEFORE sampling with the
    numbers; --write_Niter must be a multiple of --thin."""
    import os
    drv = _driver()
    common = ["--synthetic", "3,4,32", "--Nfgmodes", "3", "--seed", "4", "--out_dir", str(tmp_path), "--dry_run",
              "--write_Niter", "4", "--thin", "2", "--dirname", "t"]
    assert drv.main(common + ["--Niter", "8"]) == 0
    res = tmp_path / "dryrun-t"
    for k in (1, 2, 3):
        assert np.load(res / f"0-{k}" / "dps-eor.npy").shape == (8, 32)
        assert np.load(res / f"0-{k}" / "ln-post.npy").shape == (8,)
        assert np.load(res / f"0-{k}" / "gcr-eor.npy").shape == (4, 4, 32)
        assert np.load(res / f"0-{k}" / "fg-amps.npy").shape == (4, 4, 3)
        assert np.load(res / f"0-{k}" / "chisq.npy").shape == (4, 4, 32)
        assert np.load(res / f"0-{k}" / "cov-eor.npy").shape == (8, 32)           # periodic write: rows [:done]
    ino = os.stat(res / "0-2" / "gcr-eor.npy").st_ino
    assert drv.main(common + ["--Niter", "14", "--resume"]) == 0
    assert os.stat(res / "0-2" / "gcr-eor.npy").st_ino == ino
    assert np.load(res / "0-2" / "gcr-eor.npy").shape == (7, 4, 32) and np.load(res / "0-2" / "dps-eor.npy").shape == (14, 32)
    assert np.load(res / "0-2" / "cov-eor.npy").shape == (32, 32)                 # final write (14 % 4 != 0): full matrix
    d = json.loads((res / "drain.json").read_text())
    assert d["iter0"] == 8 and d["bytes_written"] == 3 * (6 * (32 * 8 + 8) + 3 * (4 * 32 * 16 + 4 * 3 * 16 + 4 * 32 * 8))
    with pytest.raises(SystemExit, match="thin"):
        drv.main(common + ["--Niter", "20", "--resume", "--thin", "4"])           # another thinning: another run
    with pytest.raises(SystemExit, match="pinned host memory"):
        drv.main(common + ["--Niter", "8", "--dirname", "big", "--host_mem_gb", "1e-6"])
    assert not (tmp_path / "dryrun-big" / "0-1").exists()                         # stopped before anything was written
    with pytest.raises(SystemExit, match="multiple"):
        drv.main(common + ["--Niter", "8", "--dirname", "odd", "--write_Niter", "3"])


def test_driver_drain_failure_stops_the_run(tmp_path, monkeypatch):
    """A writer-thread failure (disk full, directory gone) must not be swallowed: the sampler stops at its next flush
    with the writer's own exception, and what reached the disk before is still a valid prefix."""
    from hydra_pspec_amd import npy_append
    drv = _driver()
    real, calls = npy_append.NpyAppender.append, []

    def failing(self, rows):
        calls.append(self.path)
        if len(calls) > 6:                        # 3 baselines x 2 files of the first flush go through
            raise OSError(28, "No space left on device (simulated)")
        return real(self, rows)
    monkeypatch.setattr(npy_append.NpyAppender, "append", failing)
    with pytest.raises(OSError, match="simulated"):
        drv.main(["--synthetic", "3,4,32", "--Nfgmodes", "3", "--seed", "4", "--out_dir", str(tmp_path), "--dry_run",
                  "--write_Niter", "2", "--Niter", "8", "--outputs", "ps", "--dirname", "f"])
    monkeypatch.setattr(npy_append.NpyAppender, "append", real)
    for k in (1, 2, 3):
        assert np.load(tmp_path / "dryrun-f" / f"0-{k}" / "dps-eor.npy").shape == (2, 32)
    assert not (tmp_path / "dryrun-f" / "timings.json").exists()


@pytest.mark.gpu
def test_driver_thin_outputs_are_rows_of_the_full_run(tmp_path):
    """--thin K on the GPU: gcr-eor / fg-amps / chisq hold iterations 0, K, 2K, ... of the unthinned run bit for bit,
    dps-eor / ln-post every iteration; a --resume across a chunk boundary continues both kinds of file."""
    drv = _driver()
    common = ["--synthetic", "2,8,32", "--Nfgmodes", "4", "--ps_prior_lo", "0.1", "--ps_prior_hi", "2", "--seed", "5",
              "--out_dir", str(tmp_path), "--write_Niter", "4"]
    assert drv.main(common + ["--Niter", "12", "--dirname", "full"]) == 0
    assert drv.main(common + ["--Niter", "8", "--dirname", "thin", "--thin", "2"]) == 0
    assert drv.main(common + ["--Niter", "12", "--dirname", "thin", "--thin", "2", "--resume"]) == 0
    for k in (1, 2):
        full, thin = tmp_path / "full" / f"0-{k}", tmp_path / "thin" / f"0-{k}"
        for f in ("dps-eor.npy", "ln-post.npy", "cov-eor.npy"):
            assert np.array_equal(np.load(thin / f), np.load(full / f)), f
        for f in ("gcr-eor.npy", "fg-amps.npy", "chisq.npy"):
            assert np.array_equal(np.load(thin / f), np.load(full / f)[::2]), f


def _spawn_ranks(world, argv, env_extra=None, cwd=None):
    """Launch the driver as `world` plain processes (RANK / WORLD_SIZE / LOCAL_RANK, one shared run id), the
    way a launcher without MPI would; returns their exit codes and outputs."""
    import os, subprocess, sys, uuid
    root = Path(__file__).resolve().parent.parent
    run_id = uuid.uuid4().hex
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), HYDRA_PSPEC_RUN_ID=run_id,
                   **(env_extra or {}))
        procs.append(subprocess.Popen([sys.executable, str(root / "run-hydra-pspec.py")] + argv, env=env, cwd=cwd or root,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    return [p.returncode for p in procs], outs


def test_driver_two_ranks_dry_run(tmp_path):
    """The driver's multi-rank plumbing without a GPU (--dry_run): rank 0 prepares the output tree before the
    other rank touches it, every rank writes the directories of ITS block of baselines (the reference's quot/rem
    split, run-hydra-pspec.py:268-287), the ranks' write times are merged into one timings.json as the reference's
    gather does (:557, :570-581), --resume takes the earlier arguments from rank 0 (no rank reads args.json while
    it is rewritten) and an earlier tree is moved aside once."""
    import json
    common = ["--synthetic", "5,4,32", "--Nfgmodes", "3", "--seed", "4", "--out_dir", str(tmp_path), "--dry_run",
              "--outputs", "ps", "--write_Niter", "2"]
    rcs, outs = _spawn_ranks(2, common + ["--Niter", "4", "--dirname", "run"])
    assert rcs == [0, 0], outs
    res = tmp_path / "dryrun-run"                                    # a dry run never shares a tree with real results
    assert not (tmp_path / "run").exists()
    for k in range(1, 6):                                            # 5 baselines: rank 0 owns 3, rank 1 owns 2
        assert np.load(res / f"0-{k}" / "dps-eor.npy").shape == (4, 32)
    tm = json.load(open(res / "timings.json"))
    assert tm["num_ranks"] == 2 and tm["num_baselines"] == 5
    assert sorted(w["rank"] for w in tm["write_data"]) == [0, 1]
    assert sorted(len(w["ant_pairs"]) for w in tm["write_data"]) == [2, 3]
    assert json.load(open(res / "args.json"))["Niter"] == 4
    assert not [p for p in res.iterdir() if p.name.startswith(".timings-")]
    assert not [p for p in tmp_path.iterdir() if p.name.endswith(".ranks.json")]
    # resume on both ranks: 4 -> 6 iterations, the earlier arguments come through rank 0
    rcs, outs = _spawn_ranks(2, common + ["--Niter", "6", "--dirname", "run", "--resume"])
    assert rcs == [0, 0], outs
    for k in range(1, 6):
        assert np.load(res / f"0-{k}" / "dps-eor.npy").shape == (6, 32)
    # a resume that cannot be honoured says so instead of restarting at 0
    rcs, outs = _spawn_ranks(1, common + ["--Niter", "6", "--dirname", "run", "--resume"])
    assert rcs[0] != 0 and "nothing to do" in outs[0]
    rcs, outs = _spawn_ranks(1, common + ["--Niter", "8", "--dirname", "run", "--resume", "--seed", "9"])
    assert rcs[0] != 0 and "different" in outs[0]
    # a second plain run moves the first tree aside (rank 0, before rank 1 proceeds) and writes a fresh one
    rcs, outs = _spawn_ranks(2, common + ["--Niter", "2", "--dirname", "run"])
    assert rcs == [0, 0], outs
    moved = [p for p in tmp_path.iterdir() if p.name.startswith("dryrun-run-")]
    assert len(moved) == 1 and np.load(moved[0] / "0-5" / "dps-eor.npy").shape == (6, 32)
    assert np.load(res / "0-5" / "dps-eor.npy").shape == (2, 32) and np.load(res / "0-1" / "dps-eor.npy").shape == (2, 32)
