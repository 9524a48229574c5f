#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REAL REFERENCE.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py [--only small,steps,chain_synth,chain_testdata,chain_fullsize,steps_dense,dpss_control,chain_long [--case c3,c3f,c5f]]

The reference is imported unmodified; its two I/O-only dependencies that are
absent here (pyuvdata, astropy -- used by file loaders, never by the Gibbs path)
are registered as empty stand-in modules first.  ``hydra_pspec.oqe``
additionally needs the names ``os``/``sp``/``time`` that it forgets to import and a
``Qs/`` cache directory in the CWD; both are provided from here.  Only plain
arrays (inputs + the reference's outputs) are written -- no reference code.
"""
import argparse
import contextlib
import os
import subprocess
import sys
import tempfile
import time
import types
from pathlib import Path

import numpy as np
import scipy
import scipy.sparse.linalg

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
REF = Path("/root/reference")
sys.path.insert(0, str(REPO))


def import_reference():
    for name in ("pyuvdata", "pyuvdata.utils", "astropy", "astropy.units"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["pyuvdata"].UVData = object
    sys.modules["astropy.units"].Quantity = object
    sys.modules["astropy"].units = sys.modules["astropy.units"]
    sys.path.insert(0, str(REF))
    import hydra_pspec as hp
    assert Path(hp.__file__).parent == REF / "hydra_pspec"
    hp.oqe.os, hp.oqe.sp, hp.oqe.time = os, scipy, time
    return hp


@contextlib.contextmanager
def exact_solver():
    """Swap scipy's CG for a dense solve (the T2 control chain)."""
    orig = scipy.sparse.linalg.cg

    def direct(A, b, **kw):
        return np.linalg.solve(A, np.asarray(b).reshape(-1)), 0
    scipy.sparse.linalg.cg = direct
    try:
        yield
    finally:
        scipy.sparse.linalg.cg = orig


def load_testdata_vis():
    """pI = XX + YY of test_data/vis-eor-fgs.uvh5 for ant pair (0,1)
    (run-hydra-pspec.py:317-322, utils.py:123-130), read with the conda h5py."""
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / "d.npy"
        code = (
            "import h5py, numpy as np\n"
            f"f = h5py.File('{REF}/test_data/vis-eor-fgs.uvh5', 'r')\n"
            "pol = list(f['Header/polarization_array'][:])\n"
            "v = f['Data/visdata'][:]\n"
            "assert (f['Header/ant_1_array'][:] == 0).all() and (f['Header/ant_2_array'][:] == 1).all()\n"
            "assert not f['Data/flags'][:].any()\n"
            f"np.save('{out}', v[:, :, pol.index(-5)] + v[:, :, pol.index(-6)])\n"
        )
        subprocess.run(["/opt/conda/bin/python3.9", "-c", code], check=True)
        return np.load(out)


# --------------------------------------------------------------------------- small
def gen_small(hp):
    from hydra_pspec_amd import synthetic
    out = {}
    for n in (4, 5, 8):
        out[f"F1_fop_{n}"] = hp.utils.fourier_operator(n)
    ps = np.linspace(0.5, 2.0, 8)
    out["F2_ps"] = ps
    out["F2_cov"] = hp.pspec.covariance_from_pspec(ps, hp.utils.fourier_operator(8))
    # F3: truncated inverse-gamma draws
    cases = []
    for a in (8.0, 32.0, 203.0):
        for bscale in (0.5, 1.0, 3.0):
            for lo, hi in ((0.1, 2.0), (0.3, 5.0)):
                for k in range(3):
                    seed = 100 + k
                    beta = a * bscale
                    np.random.seed(seed)
                    v = float(hp.pspec.inversion_sample_invgamma(a, beta, lo, hi))
                    u_after = np.random.uniform()
                    cases.append((a, beta, lo, hi, seed, v, u_after))
    out["F3_cases"] = np.array(cases)
    # F4: sample_S
    rng = np.random.default_rng(5)
    s = (rng.standard_normal((8, 16)) + 1j * rng.standard_normal((8, 16)))
    pr = np.zeros((2, 16)); pr[0, 6:11] = 400.0; pr[1, 6:11] = 20.0
    out["F4_s"], out["F4_prior"] = s, pr
    np.random.seed(11); out["F4_x_noprior"] = hp.pspec.sample_S(s=s)
    np.random.seed(11); out["F4_x_prior"] = hp.pspec.sample_S(s=s, prior=pr)
    out["F4_next_uniform"] = np.random.uniform()
    # F5: sprior
    out["F5_prior"] = hp.pspec.sprior(s, 2, 10.0)
    # F6/F7: build_matrices + gcr_fgmodes_1d
    for tag, frac in (("nf", 0.0), ("fl", 0.2)):
        d = synthetic.make_baselines(16, 4, 3, k0=50, flag_frac=frac)
        fl = d["flags"][0]
        mats = hp.pspec.build_matrices(19, fl, d["S_initial"], d["Ninv"], d["fgmodes"])
        for k in ("vis", "flags", "S_initial", "Ninv", "fgmodes"):
            out[f"F6_{tag}_{k}"] = d[k][0] if k in ("vis", "flags") else d[k]
        out[f"F6_{tag}_ops"], out[f"F6_{tag}_sys"] = mats[0], mats[1]
        xs = []
        for idx in (0, 3):
            x, _, info = hp.pspec.gcr_fgmodes_1d(idx, (d["vis"][0] * fl)[idx], fl, mats, d["fgmodes"])
            assert info == 0
            xs.append(x)
        out[f"F7_{tag}_x"] = np.array(xs)
        x, _, _ = hp.pspec.gcr_fgmodes_1d(1, (d["vis"][0] * fl)[1], fl, mats, d["fgmodes"], map_estimate=True)
        out[f"F7_{tag}_xmap"] = x
    # omega stream itself (seed 912983+idx, 4 x randn(N,1))
    np.random.seed(912983 + 2)
    out["F7_omega_idx2_N6"] = np.array([np.random.randn(6, 1)[:, 0] for _ in range(4)])
    # F10: dpss_fit_modes
    for i, (n, nm, al) in enumerate(((32, 4, 2.0), (48, 6, 3.0), (64, 6, 3.0))):
        rng = np.random.default_rng(77 + i)
        freqs = np.linspace(100., 120., n)
        d = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        w = np.ones(n); w[rng.choice(n, n // 8, replace=False)] = 0.0
        a = rng.standard_normal((n, n)); cov = a @ a.T / n + np.eye(n)
        taper = np.hanning(n + 2)[1:-1] if i == 2 else None
        modes, amps = hp.dpss.dpss_fit_modes(d, w, freqs, cov, nmodes=nm, alpha=al, taper=taper)
        out[f"F10_{i}_d"], out[f"F10_{i}_w"], out[f"F10_{i}_freqs"], out[f"F10_{i}_cov"] = d, w, freqs, cov
        out[f"F10_{i}_par"] = np.array([nm, al, 1.0 if taper is not None else 0.0])
        if taper is not None:
            out[f"F10_{i}_taper"] = taper
        out[f"F10_{i}_modes"], out[f"F10_{i}_amps"] = modes, amps
    # F11: oqe
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td); os.mkdir("Qs")
        try:
            for s_ in (8, 16):
                rng = np.random.default_rng(s_)
                a = rng.standard_normal((s_, s_)) + 1j * rng.standard_normal((s_, s_))
                R = a @ a.conj().T / s_ + np.eye(s_)            # Hermitian weighting
                Rg = a / np.sqrt(s_) + np.eye(s_)                # general (non-Hermitian)
                V = rng.standard_normal((6, s_)) + 1j * rng.standard_normal((6, s_))
                Cn = np.diag(rng.uniform(0.5, 1.5, s_)).astype(complex)
                out[f"F11_{s_}_R"], out[f"F11_{s_}_Rg"], out[f"F11_{s_}_V"], out[f"F11_{s_}_Cn"] = R, Rg, V, Cn
                out[f"F11_{s_}_Q3"] = hp.oqe.Q(3, s_)
                Fm = hp.oqe.F(s_, R)
                out[f"F11_{s_}_F"] = Fm
                out[f"F11_{s_}_Ft"] = hp.oqe.Ft(s_, R)
                out[f"F11_{s_}_Fg"] = hp.oqe.F(s_, Rg)
                out[f"F11_{s_}_Ftg"] = hp.oqe.Ft(s_, Rg)
                out[f"F11_{s_}_Mopt"] = hp.oqe.M_opt(Fm)
                out[f"F11_{s_}_MFinv"] = hp.oqe.M_Finv(Fm)
                out[f"F11_{s_}_MFhalf"] = hp.oqe.M_Fhalf(Fm)
                out[f"F11_{s_}_qh"] = hp.oqe.q_h(V, s_, R)
                out[f"F11_{s_}_qhg"] = hp.oqe.q_h(V, s_, Rg)
                b = np.array([hp.oqe.bias(t, s_, R, Cn) for t in range(s_)])
                out[f"F11_{s_}_bias"] = b
                out[f"F11_{s_}_q"] = hp.oqe.q(V, s_, R, b.real)
                out[f"F11_{s_}_SigN"] = hp.oqe.Sig_QEN(R, Cn, 0.37)
                out[f"F11_{s_}_SigSN"] = hp.oqe.Sig_QESN(R, Cn, R, 0.37)
        finally:
            os.chdir(cwd)
    np.savez(HERE / "small.npz", **out)
    print("small.npz", len(out), "arrays")


# --------------------------------------------------------------------------- steps
STEP_CASES = [
    # name, T, N, M, flag_frac, prior, S kind
    ("a", 8, 32, 4, 0.0, True, "true"),
    ("b", 8, 32, 4, 0.15, True, "true"),
    ("c", 8, 32, 4, 0.0, False, "true"),
    ("d", 8, 32, 4, 0.15, False, "true"),
    ("e", 8, 64, 6, 0.0, True, "true"),
    ("f", 8, 64, 6, 0.15, True, "true"),
    ("g", 8, 64, 6, 0.0, False, "eye"),
    ("h", 8, 64, 6, 0.15, True, "general"),
    ("i", 6, 30, 5, 0.1, True, "true"),      # non power of two
]


def step_inputs(name, T, N, M, frac, prior, skind):
    from hydra_pspec_amd import synthetic
    d = synthetic.make_baselines(N, T, M, k0=200 + ord(name), flag_frac=frac, prior=prior)
    if skind == "eye":
        S = np.eye(N)
    elif skind == "general":
        rng = np.random.default_rng(9)
        a = (rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))) / np.sqrt(N)
        S = d["S_initial"] + 0.3 * np.trace(d["S_initial"]).real / N * (a @ a.conj().T)
        S = 0.5 * (S + S.conj().T)
    else:
        S = d["S_initial"]
    return dict(vis=d["vis"][0], flags=d["flags"][0], S=S, fgmodes=d["fgmodes"],
                Ninv=d["Ninv"], prior=d["ps_prior"])


def gen_steps(hp):
    out = {}
    for case in STEP_CASES:
        name = case[0]
        inp = step_inputs(*case)
        np.random.seed(4242)
        cr, S_s, ps, fg, chi, lp = hp.pspec.gibbs_step_fgmodes(
            vis=inp["vis"] * inp["flags"], flags=inp["flags"], signal_S=inp["S"],
            fgmodes=inp["fgmodes"], Ninv=inp["Ninv"], ps_prior=inp["prior"], nproc=1)
        for k, v in inp.items():
            out[f"{name}_in_{k}"] = v
        out[f"{name}_cr"], out[f"{name}_S"], out[f"{name}_ps"] = cr, S_s, ps
        out[f"{name}_fg"], out[f"{name}_chisq"], out[f"{name}_lnpost"] = fg, chi, np.array(lp)
        print("step", name, "ps[:3]", ps[:3], "lnpost", lp)
    # F12 map_estimate through the chain entry
    inp = step_inputs(*STEP_CASES[1])
    np.random.seed(3)   # map_estimate does not reseed: the draw uses the caller's stream
    res = hp.pspec.gibbs_sample_with_fg(inp["vis"], inp["flags"], inp["S"], inp["fgmodes"], inp["Ninv"],
                                        inp["prior"], Niter=5, seed=3, verbose=False, nproc=1,
                                        map_estimate=True)
    out["map_cr"], out["map_S"], out["map_ps"], out["map_fg"] = res[0], res[1], res[2], res[3]
    out["map_chisq"], out["map_lnpost"] = res[4], res[5]
    np.savez(HERE / "steps.npz", **out)
    print("steps.npz", len(out), "arrays")


# --------------------------------------------------------------------------- dense noise covariance
def dense_noise_inputs():
    """A banded, complex Hermitian positive-definite noise covariance (neighbouring channels
    correlated) and unflagged data: the case in which the reference's Ni = flags.T * Ninv * flags stays
    Hermitian (pspec.py:361).  Ninv = inv(noise_cov) as the driver forms it (run-hydra-pspec.py:436)."""
    from hydra_pspec_amd import synthetic
    T, N, M = 8, 32, 4
    d = synthetic.make_baselines(N, T, M, k0=333, flag_frac=0.0, prior=True)
    sig2 = 1.0 / d["Ninv"][0, 0].real
    i = np.arange(N)
    band = np.zeros((N, N), dtype=complex)
    band[i, i] = 1.0 + 0.2 * np.cos(0.3 * i)
    band[i[:-1], i[:-1] + 1] = 0.3 * np.exp(0.4j)
    band[i[:-1] + 1, i[:-1]] = 0.3 * np.exp(-0.4j)
    band[i[:-2], i[:-2] + 2] = 0.1
    band[i[:-2] + 2, i[:-2]] = 0.1
    noise_cov = sig2 * band
    return dict(vis=d["vis"][0], flags=d["flags"][0], S=d["S_initial"], fgmodes=d["fgmodes"],
                Ninv=np.linalg.inv(noise_cov), prior=d["ps_prior"], noise_cov=noise_cov)


def gen_steps_dense(hp):
    import warnings
    out = {}
    inp = dense_noise_inputs()
    for k, v in inp.items():
        out[f"in_{k}"] = v
    np.random.seed(4242)
    cr, S_s, ps, fg, chi, lp = hp.pspec.gibbs_step_fgmodes(
        vis=inp["vis"] * inp["flags"], flags=inp["flags"], signal_S=inp["S"], fgmodes=inp["fgmodes"],
        Ninv=inp["Ninv"], ps_prior=inp["prior"], nproc=1)
    out["step_cr"], out["step_S"], out["step_ps"], out["step_fg"] = cr, S_s, ps, fg
    out["step_chisq"], out["step_lnpost"] = chi, np.array(lp)          # chisq is complex here (Ninv.diagonal())
    print("dense step ps[:3]", ps[:3], "lnpost", lp)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")      # the chain driver stores the complex chi^2 into a real array
        r = run_chain(hp, inp["vis"], inp["flags"], inp["S"], inp["fgmodes"], inp["Ninv"], inp["prior"], 6, 77)
        pack_chain(out, "chain_", r)
        with exact_solver():
            r2 = run_chain(hp, inp["vis"], inp["flags"], inp["S"], inp["fgmodes"], inp["Ninv"], inp["prior"], 6, 77)
        out["chain_exact_ps"], out["chain_exact_lnpost"] = r2[2], r2[5]
    # the same noise covariance WITH flagged channels: the reference's column-masked Ni = flags.T * Ninv * flags
    # is not Hermitian there (pspec.py:361 FIXME); it runs -- sqrtm of the masked matrix, CG on the non-Hermitian
    # system preconditioned with its pseudo-inverse -- and this is what it returns
    flags = inp["flags"].copy()
    flags[np.random.default_rng(5).choice(flags.size, size=5, replace=False)] = False
    out["fl_flags"] = flags
    np.random.seed(4242)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cr, S_s, ps, fg, chi, lp = hp.pspec.gibbs_step_fgmodes(
            vis=inp["vis"] * flags, flags=flags, signal_S=inp["S"], fgmodes=inp["fgmodes"],
            Ninv=inp["Ninv"], ps_prior=inp["prior"], nproc=1)
        out["fl_step_cr"], out["fl_step_ps"], out["fl_step_fg"] = cr, ps, fg
        out["fl_step_chisq"], out["fl_step_lnpost"] = chi, np.array(lp)
        print("dense + flags step ps[:3]", ps[:3], "lnpost", lp)
        r = run_chain(hp, inp["vis"], flags, inp["S"], inp["fgmodes"], inp["Ninv"], inp["prior"], 6, 77)
        pack_chain(out, "fl_chain_", r)
        with exact_solver():
            r2 = run_chain(hp, inp["vis"], flags, inp["S"], inp["fgmodes"], inp["Ninv"], inp["prior"], 6, 77)
        out["fl_chain_exact_ps"], out["fl_chain_exact_lnpost"] = r2[2], r2[5]
    np.savez(HERE / "steps_dense.npz", **out)
    print("steps_dense.npz", len(out), "arrays")


# --------------------------------------------------------------------------- chains
def run_chain(hp, vis, flags, S0, F, Ninv, prior, niter, seed):
    t0 = time.time()
    r = hp.pspec.gibbs_sample_with_fg(vis, flags, S0, F, Ninv, prior, Niter=niter, seed=seed,
                                      verbose=False, nproc=1, out_dir=None)
    print(f"   chain {niter} iters: {time.time() - t0:.1f}s")
    return r


def pack_chain(out, pre, r):
    cr, S_last, ps, fg, chi, lp, _ = r
    out[pre + "ps"], out[pre + "lnpost"] = ps, lp
    out[pre + "S_last"] = S_last
    sel = sorted({0, 1, len(ps) // 2, len(ps) - 1})
    out[pre + "sel"] = np.array(sel)
    out[pre + "cr_sel"], out[pre + "fg_sel"], out[pre + "chisq_sel"] = cr[sel], fg[sel], chi[sel]


def gen_chain_synth(hp):
    from hydra_pspec_amd import synthetic
    out = {}
    T, N, M, niter = 16, 64, 6, 200
    for b, frac in enumerate((0.0, 0.0, 0.15)):
        d = synthetic.make_baselines(N, T, M, k0=b, flag_frac=frac)
        vis, fl = d["vis"][0], d["flags"][0]
        out[f"b{b}_vis"], out[f"b{b}_flags"] = vis, fl
        pack_chain(out, f"b{b}_ref_", run_chain(hp, vis, fl, d["S_initial"], d["fgmodes"], d["Ninv"],
                                                d["ps_prior"], niter, d["seed"]))
        with exact_solver():
            r = run_chain(hp, vis, fl, d["S_initial"], d["fgmodes"], d["Ninv"], d["ps_prior"], niter, d["seed"])
        out[f"b{b}_exact_ps"], out[f"b{b}_exact_lnpost"] = r[2], r[5]
    out["S_initial"], out["fgmodes"], out["Ninv"], out["prior"] = d["S_initial"], d["fgmodes"], d["Ninv"], d["ps_prior"]
    out["seed"] = np.array(d["seed"])
    np.savez(HERE / "chain_synth.npz", **out)
    print("chain_synth.npz", len(out), "arrays")


def gen_chain_testdata(hp):
    """BASELINE.json config 1: test_data, ant pair 0-1, 200 iterations."""
    out = {}
    td = REF / "test_data" / "0-1"
    d = load_testdata_vis() + np.load(td / "noise.npy")           # run-hydra-pspec.py:415
    Ninv = np.linalg.inv(np.load(td / "noise-cov.npy"))           # :436
    F = np.load(td / "fgmodes.npy")[:, :12]                       # :453
    S0 = np.load(td / "eor-cov.npy")
    N = d.shape[1]
    prior = np.zeros((2, N)); prior[0, N // 2 - 3:N // 2 + 4] = 2.0; prior[1, N // 2 - 3:N // 2 + 4] = 0.1
    flags = np.ones(N, dtype=bool)
    niter, seed = 200, 7123689
    out.update(vis=d, flags=flags, S_initial=S0, fgmodes=F, ninv_diag=np.diag(Ninv).copy(), prior=prior,
               seed=np.array(seed))
    assert np.allclose(Ninv, np.diag(np.diag(Ninv)))
    pack_chain(out, "ref_", run_chain(hp, d, flags, S0, F, Ninv, prior, niter, seed))
    with exact_solver():
        r = run_chain(hp, d, flags, S0, F, Ninv, prior, niter, seed)
    out["exact_ps"], out["exact_lnpost"] = r[2], r[5]
    # true EoR delay spectrum for the recovery check is not stored (needs vis-eor.uvh5);
    # the statistical test compares against the reference chain's own posterior.
    np.savez(HERE / "chain_testdata.npz", **out)
    print("chain_testdata.npz", len(out), "arrays")


def gen_chain_fullsize(hp):
    """The reference itself at BASELINE.json's full channel counts: the C3 shape (32, 512, 12) without
    and with 15 % flags, and the C5 shape (32, 1024, 12) with 15 % flags; a few iterations each
    (1.7 s / 12 s per iteration here).  Inputs are stored with the outputs."""
    from hydra_pspec_amd import synthetic
    out = {}
    cases = (("c3", 512, 0.0, 40, 4), ("c3f", 512, 0.15, 41, 4), ("c5f", 1024, 0.15, 42, 3))
    for tag, N, frac, k0, niter in cases:
        T, M = 32, 12
        d = synthetic.make_baselines(N, T, M, k0=k0, flag_frac=frac)
        vis, fl = d["vis"][0], d["flags"][0]
        r = run_chain(hp, vis, fl, d["S_initial"], d["fgmodes"], d["Ninv"], d["ps_prior"], niter, d["seed"])
        cr, S_last, ps, fg, chi, lp, _ = r
        out[f"{tag}_vis"], out[f"{tag}_flags"] = vis, fl
        out[f"{tag}_fgmodes"], out[f"{tag}_ninv_diag"] = d["fgmodes"], np.diag(d["Ninv"]).real.copy()
        out[f"{tag}_prior"], out[f"{tag}_ps0"], out[f"{tag}_seed"] = d["ps_prior"], d["ps0"], np.array(d["seed"])
        out[f"{tag}_ps"], out[f"{tag}_lnpost"] = ps, lp
        out[f"{tag}_cr_last"], out[f"{tag}_fg"], out[f"{tag}_chisq_last"] = cr[-1], fg, chi[-1]
        print("fullsize", tag, "lnpost", lp)
    np.savez_compressed(HERE / "chain_fullsize.npz", **out)
    print("chain_fullsize.npz", len(out), "arrays")


def gen_dpss_control(hp):
    """D1 control (the device exact_solver() is for the chain): the reference's dpss_fit_modes on the F10 inputs of
    small.npz with ITS optimiser call (dpss.py:86-92: scipy.optimize.minimize, L-BFGS-B, default tolerances) swapped
    in-process for the same call with tight stopping tolerances -- what the reference converges to when it is allowed to
    converge.  SURVEY 8(a) D1 gates the closed form against this at 1e-6 of max |c|."""
    import scipy.optimize
    g = dict(np.load(HERE / "small.npz"))
    out = {}
    orig = hp.dpss.minimize

    def tight(fun, x0, method=None, bounds=None, **kw):
        return scipy.optimize.minimize(fun, x0, method=method, bounds=bounds,
                                       options={"ftol": 1e-15, "gtol": 1e-12, "maxiter": 100000, "maxfun": 10000000})
    hp.dpss.minimize = tight
    try:
        for i in range(3):
            nm, al, has_t = g[f"F10_{i}_par"]
            taper = g[f"F10_{i}_taper"] if has_t else None
            modes, amps = hp.dpss.dpss_fit_modes(g[f"F10_{i}_d"], g[f"F10_{i}_w"], g[f"F10_{i}_freqs"], g[f"F10_{i}_cov"],
                                                 nmodes=int(nm), alpha=al, taper=taper)
            assert np.array_equal(modes, g[f"F10_{i}_modes"])
            out[f"F10_{i}_amps_tight"] = amps
            print("dpss control", i, "max |tight - default| / max|c| =",
                  np.max(np.abs(amps - g[f"F10_{i}_amps"])) / np.max(np.abs(amps)))
    finally:
        hp.dpss.minimize = orig
    np.savez(HERE / "dpss_control.npz", **out)
    print("dpss_control.npz", len(out), "arrays")


LONG_CASES = {"c3": (512, 0.0, 40, 200), "c3f": (512, 0.15, 41, 100), "c5f": (1024, 0.15, 42, 30)}


def gen_chain_long(hp, cases):
    """The reference AND its exact-solve control, free-running, long enough for SURVEY 8c's T2 gates at
    BASELINE.json's channel counts: 200 iterations at (32, 512, 12) without flags, 100 with 15 % flags,
    30 at (32, 1024, 12) with 15 % flags.  Same inputs as chain_fullsize.npz (same k0), one file per case
    (chain_long_<tag>.npz) so that the cases can be produced by separate processes."""
    from hydra_pspec_amd import synthetic
    for tag in cases:
        N, frac, k0, niter = LONG_CASES[tag]
        T, M = 32, 12
        d = synthetic.make_baselines(N, T, M, k0=k0, flag_frac=frac)
        vis, fl = d["vis"][0], d["flags"][0]
        out = dict(vis=vis, flags=fl, fgmodes=d["fgmodes"], ninv_diag=np.diag(d["Ninv"]).real.copy(),
                   prior=d["ps_prior"], ps0=d["ps0"], seed=np.array(d["seed"]))
        r = run_chain(hp, vis, fl, d["S_initial"], d["fgmodes"], d["Ninv"], d["ps_prior"], niter, d["seed"])
        cr, _, ps, fg, chi, lp, _ = r
        sel = sorted({0, 1, niter // 2, niter - 1})
        out.update(ref_ps=ps, ref_lnpost=lp, sel=np.array(sel), ref_fg_sel=fg[sel],
                   ref_cr_sel=cr[sel].astype(np.complex128), ref_chisq_sel=chi[sel])
        with exact_solver():
            r = run_chain(hp, vis, fl, d["S_initial"], d["fgmodes"], d["Ninv"], d["ps_prior"], niter, d["seed"])
        out.update(exact_ps=r[2], exact_lnpost=r[5])
        np.savez_compressed(HERE / f"chain_long_{tag}.npz", **out)
        print(f"chain_long_{tag}.npz", len(out), "arrays")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="small,steps,chain_synth,chain_testdata")
    ap.add_argument("--case", default="c3,c3f,c5f", help="cases of chain_long")
    args = ap.parse_args()
    hp = import_reference()
    todo = args.only.split(",")
    if "small" in todo:
        gen_small(hp)
    if "steps" in todo:
        gen_steps(hp)
    if "chain_synth" in todo:
        gen_chain_synth(hp)
    if "chain_testdata" in todo:
        gen_chain_testdata(hp)
    if "chain_fullsize" in todo:
        gen_chain_fullsize(hp)
    if "steps_dense" in todo:
        gen_steps_dense(hp)
    if "dpss_control" in todo:
        gen_dpss_control(hp)
    if "chain_long" in todo:
        gen_chain_long(hp, args.case.split(","))


if __name__ == "__main__":
    main()
