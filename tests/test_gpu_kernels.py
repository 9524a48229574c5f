"""Unit parity of the HIP stages against numpy, through the C-ABI (GPU box only)."""
import ctypes as C

import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch
    from hydra_pspec_amd import hpx
    hpx.require_gpu()
    return torch


def _dev(torch, x, dtype):
    return torch.from_numpy(np.ascontiguousarray(x)).to("cuda", dtype=dtype).contiguous()


def test_mfma_f64_lane_map():
    """The accumulator lane map the kernels assume (HPX_ACC_ROW(g,v) = g + 4v)."""
    from hydra_pspec_amd import hpx
    rng = np.random.default_rng(0)
    A = rng.integers(-9, 10, size=(16, 4)).astype(float)
    B = rng.integers(-9, 10, size=(4, 16)).astype(float)       # asymmetric
    D = np.zeros((64, 4))
    hpx.check(hpx.lib().hpx_mfma_probe(A.ctypes.data_as(C.c_void_p), B.ctypes.data_as(C.c_void_p),
                                       D.ctypes.data_as(C.c_void_p)))
    ref = A @ B
    for lane in range(64):
        for v in range(4):
            assert D[lane, v] == ref[(lane >> 4) + 4 * v, lane & 15], (lane, v)


def _hpd(rng, nb, n, cond=1e3):
    a = rng.standard_normal((nb, n, n)) + 1j * rng.standard_normal((nb, n, n))
    q, _ = np.linalg.qr(a)
    ev = np.logspace(0, np.log10(cond), n)
    return (q * ev[None, None, :]) @ np.conj(np.swapaxes(q, 1, 2))


@pytest.mark.parametrize("n", [5, 16, 35, 64, 132, 400, 524, 652, 1036])
def test_zpotrf(T, n):
    from hydra_pspec_amd import hpx
    rng = np.random.default_rng(n)
    nb = 3
    A = _hpd(rng, nb, n)
    A = 0.5 * (A + np.conj(np.swapaxes(A, 1, 2)))
    dA = _dev(T, A, T.complex128)
    dL = T.zeros_like(dA)
    info = T.zeros(nb, dtype=T.int32, device="cuda")
    hpx.check(hpx.lib().hpx_zpotrf_batched(nb, n, hpx.ptr(dA), hpx.ptr(dL), hpx.ptr(info), None))
    L = dL.cpu().numpy()
    assert not info.cpu().numpy().any()
    ref = np.linalg.cholesky(A)
    assert relerr(L, ref) < 1e-11
    assert relerr(L @ np.conj(np.swapaxes(L, 1, 2)), A) < 1e-13


def test_zpotrf_not_positive_definite(T):
    from hydra_pspec_amd import hpx
    A = np.eye(20, dtype=complex)[None].repeat(2, 0)
    A[1, 7, 7] = -1.0
    dA = _dev(T, A, T.complex128)
    dL = T.zeros_like(dA)
    info = T.zeros(2, dtype=T.int32, device="cuda")
    hpx.check(hpx.lib().hpx_zpotrf_batched(2, 20, hpx.ptr(dA), hpx.ptr(dL), hpx.ptr(info), None))
    assert info.cpu().numpy().tolist() == [0, 1]


@pytest.mark.parametrize("n,nrhs", [(16, 16), (35, 3), (132, 203), (524, 32), (150, 48), (780, 48), (513, 1)])
def test_zpotrs(T, n, nrhs):
    from hydra_pspec_amd import hpx
    rng = np.random.default_rng(n + nrhs)
    nb = 2
    A = _hpd(rng, nb, n)
    A = 0.5 * (A + np.conj(np.swapaxes(A, 1, 2)))
    B = rng.standard_normal((nb, n, nrhs)) + 1j * rng.standard_normal((nb, n, nrhs))
    dA, dB = _dev(T, A, T.complex128), _dev(T, B, T.complex128)
    dX = T.zeros_like(dB)
    info = T.zeros(nb, dtype=T.int32, device="cuda")
    hpx.check(hpx.lib().hpx_zpotrs_batched(nb, n, nrhs, hpx.ptr(dA), hpx.ptr(dB), hpx.ptr(dX),
                                           hpx.ptr(info), None))
    X = dX.cpu().numpy()
    assert relerr(X, np.linalg.solve(A, B)) < 1e-10


@pytest.mark.parametrize("N,Tn", [(8, 3), (30, 6), (120, 20), (512, 32)])
def test_dft(T, N, Tn):
    from hydra_pspec_amd import hpx, utils
    rng = np.random.default_rng(N)
    fop = utils.fourier_operator(N)
    x = rng.standard_normal((2, Tn, N)) + 1j * rng.standard_normal((2, Tn, N))
    dF, dx = _dev(T, fop, T.complex128), _dev(T, x, T.complex128)
    dy = T.zeros_like(dx)
    hpx.check(hpx.lib().hpx_dft_batched(2, Tn, N, hpx.ptr(dF), hpx.ptr(dx), hpx.ptr(dy), 0, None))
    ref = np.fft.fftshift(np.fft.fft(np.fft.ifftshift(x, axes=-1), axis=-1), axes=-1)
    assert relerr(dy.cpu().numpy(), ref) < 1e-13
    hpx.check(hpx.lib().hpx_dft_batched(2, Tn, N, hpx.ptr(dF), hpx.ptr(dy), hpx.ptr(dx), 1, None))
    assert relerr(dx.cpu().numpy(), x) < 1e-13


def test_invgamma_inversion_golden(T, golden):
    """pspec.py:11-64 against the reference's own draws (tests/golden small.npz F3)."""
    from hydra_pspec_amd import hpx
    cases = golden("small")["F3_cases"]
    n = len(cases)
    beta, u, xg, want = np.zeros(n), np.zeros(n), np.zeros((n, 1000)), np.zeros(n)
    alphas = cases[:, 0].astype(int)
    for i, (a, b, lo, hi, seed, v, _) in enumerate(cases):
        np.random.seed(int(seed))
        u[i] = np.random.uniform()
        beta[i], want[i] = b, v
        xg[i] = np.logspace(np.log10(lo), np.log10(hi), 1000)
    for a in np.unique(alphas):
        sel = np.nonzero(alphas == a)[0]
        out = T.zeros(len(sel), dtype=T.float64, device="cuda")
        db, du, dx = _dev(T, beta[sel], T.float64), _dev(T, u[sel], T.float64), _dev(T, xg[sel], T.float64)
        hpx.check(hpx.lib().hpx_invgamma_inversion(len(sel), int(a), hpx.ptr(db), hpx.ptr(du), hpx.ptr(dx),
                                                   1000, hpx.ptr(out), None))
        assert np.max(np.abs(out.cpu().numpy() / want[sel] - 1)) < 1e-10


def test_invgamma_inversion_flat_ends_and_widths(T, monkeypatch):
    """The draw's short cuts (bracket from one round of neighbouring table values; 16 / 32 / 64 lanes per draw) against
    the searches of the general form: tables with long runs of equal values at both ends (the CDF saturates over most
    of a wide grid), u from 1e-300 to 1 - 1e-13 (within a few ulp of 1 the computed table is no longer monotone where it
    saturates, and which of the equal-looking points a search lands on is its own business).  The kernel returns NaN unless the three group widths agree to the
    bit -- a run of equal values between 16 and 63 points long takes the short cut at one width and the general
    searches at another.  Where u is inside the table's resolved part the value is the oracle's (pspec.py:11-64)."""
    from hydra_pspec_amd import hpx
    from oracle import pspec_ref
    rng = np.random.default_rng(11)
    us = [1e-300, 1e-17, 1e-9, 1e-3, 0.25, 0.5, 0.9, 1 - 1e-9, 1 - 1e-13]
    cases = []
    for a in (2, 8, 32):
        for _ in range(6):
            b = 10.0 ** rng.uniform(-3, 3)
            lo = b / a * 10.0 ** rng.uniform(-5, -0.3)      # (beta / x from far above the shape to far below it:
            hi = b / a * 10.0 ** rng.uniform(0.5, 6)        # the CDF rises inside the grid, flat on both sides)
            cases += [(a, b, lo, hi, u) for u in us]
    for a in (2, 8, 32):
        sel = [c for c in cases if c[0] == a]
        beta = np.array([c[1] for c in sel])
        u = np.array([c[4] for c in sel])
        xg = np.stack([np.logspace(np.log10(c[2]), np.log10(c[3]), 1000) for c in sel])
        out = T.zeros(len(sel), dtype=T.float64, device="cuda")
        db, du, dx = _dev(T, beta, T.float64), _dev(T, u, T.float64), _dev(T, xg, T.float64)
        hpx.check(hpx.lib().hpx_invgamma_inversion(len(sel), a, hpx.ptr(db), hpx.ptr(du), hpx.ptr(dx), 1000,
                                                   hpx.ptr(out), None))
        got = out.cpu().numpy()
        assert np.isfinite(got).all(), [c for c, g in zip(sel, got) if not np.isfinite(g)]   # (NaN: the widths disagree)
        for g, (_, b, lo, hi, uu) in zip(got, sel):
            assert lo * (1 - 1e-12) <= g <= hi * (1 + 1e-12)
            if 1e-3 <= uu <= 0.9:
                monkeypatch.setattr(np.random, "uniform", lambda: uu)
                want = float(pspec_ref.inversion_sample_invgamma(a, b, lo, hi))
                assert g == pytest.approx(want, rel=1e-8)


def test_inversion_sample_invgamma_api(T, golden):
    from hydra_pspec_amd import pspec
    for a, b, lo, hi, seed, v, u_after in golden("small")["F3_cases"][::5]:
        np.random.seed(int(seed))
        got = pspec.inversion_sample_invgamma(a, b, lo, hi)
        assert got == pytest.approx(v, rel=1e-10)
        assert np.random.uniform() == u_after
    for bad in [(0.0, 1.0), (1.0, 0.0), (1.0, np.inf), (2.0, 1.0)]:
        with pytest.raises(ValueError):
            pspec.inversion_sample_invgamma(5, 5.0, *bad)


def _reference_system(vis, flags, ninv, F, ps, omega):
    """numpy construction of K' and r' from their definitions."""
    from oracle import pspec_ref as R
    Tn, N = vis.shape
    fop = R.fourier_operator(N)
    U = fop.conj().T / np.sqrt(N)
    w = flags.astype(float)
    ni = ninv * w
    nih = np.sqrt(ni)
    a = np.sqrt(ps / N)
    Cm = U.conj().T @ (ni[:, None] * U)
    G = U.conj().T @ (ni[:, None] * F)
    H = F.conj().T @ (ni[:, None] * F)
    d = vis * w
    if omega is None:
        oma = omb = np.zeros((Tn, N), dtype=complex)
    else:
        oma = (omega[:, 0] + 1j * omega[:, 1]) / 2 ** 0.5
        omb = (omega[:, 2] + 1j * omega[:, 3]) / 2 ** 0.5
    z = ni[:, None] * d.T + nih[:, None] * omb.T
    rtop = a[:, None] * (U.conj().T @ z) + U.conj().T @ oma.T
    rbot = F.conj().T @ z
    K = np.block([[np.eye(N) + a[:, None] * Cm * a[None, :], a[:, None] * G],
                  [G.conj().T * a[None, :], H]])
    return K, np.vstack([rtop, rbot]), U, a


@pytest.mark.parametrize("Tn,N,M,frac", [(8, 32, 4, 0.0), (6, 30, 5, 0.2), (20, 120, 12, 0.1)])
def test_assemble_K(T, Tn, N, M, frac):
    from hydra_pspec_amd import hpx, pspec, synthetic
    d = synthetic.make_baselines(N, Tn, M, k0=7, nbl=2, flag_frac=frac)
    F = d["fgmodes"] * (1 + 0.3j)          # exercise complex modes
    gb = pspec.GibbsBatch(d["vis"], d["flags"], F, d["ninv_diag"], d["ps_prior"], 2, seed=1)
    npad, tp, ld = gb.plan.dims()
    ps = d["ps0"] * np.linspace(0.5, 1.5, N)
    dps = _dev(T, np.stack([ps, ps[::-1]]), T.float64)
    out = T.zeros((2, ld, npad), dtype=T.complex128, device="cuda")
    hpx.check(hpx.lib().hpx_assemble_K(gb.plan.handle, hpx.ptr(dps), hpx.ptr(out), None))
    Kd = out.cpu().numpy()
    om = pspec.omega_table(Tn, N)
    n = N + M
    for b, psb in enumerate((ps, ps[::-1])):
        K, r, _, a = _reference_system(d["vis"][b], d["flags"][b], d["ninv_diag"][b], F, psb, om)
        # the device works with the symmetrically scaled system M = A^-1 K' A^-1, A = diag(a, 1)
        # (hpx_internal.h): K' entries and right-hand sides divided by a on the signal side
        ainv = np.concatenate([1.0 / a, np.ones(M)])
        K = ainv[:, None] * K * ainv[None, :]
        r = ainv[:, None] * r
        got = Kd[b]
        assert relerr(np.tril(got[:n, :n]), np.tril(K)) < 1e-12
        assert relerr(got[npad:npad + Tn, :n], r.conj().T) < 1e-12
        pad = got[n:npad, :]
        assert np.array_equal(pad[:, n:npad], np.eye(npad - n)) and not pad[:, :n].any()
    gb.close()


def test_zpotrs_beyond_the_register_form(T):
    """Orders 273 .. 1050 with 17 .. 32 right-hand sides (32 columns) and with 1 .. 16 (16 columns): the back substitution of
    csrc/hpx_backsolve_lds.hip (eight waves, the solution rows through an LDS ring, hand-counted operand waits) --
    tile counts around the super-block boundaries of 8 tiles (128 columns), one to several groups of four chunks per
    pass, a top super-block of one tile (C3's 33) and of eight, against numpy."""
    from hydra_pspec_amd import hpx
    rng = np.random.default_rng(2024)
    sizes = [273, 288, 300, 383, 384, 385, 400, 496, 512, 513, 524, 528, 529, 640, 777, 1040, 1050]
    worst = 0.0
    for i, n in enumerate(sizes):
        nrhs = [32, 17, 1, 24, 16, 31, 9][i % 7]
        nb = 1 + i % 3
        A = _hpd(rng, nb, n, cond=1e3)
        A = 0.5 * (A + np.conj(np.swapaxes(A, 1, 2)))
        B = rng.standard_normal((nb, n, nrhs)) + 1j * rng.standard_normal((nb, n, nrhs))
        dA, dB = _dev(T, A, T.complex128), _dev(T, B, T.complex128)
        dX = T.zeros_like(dB)
        info = T.zeros(nb, dtype=T.int32, device="cuda")
        hpx.check(hpx.lib().hpx_zpotrs_batched(nb, n, nrhs, hpx.ptr(dA), hpx.ptr(dB), hpx.ptr(dX),
                                               hpx.ptr(info), None))
        assert not info.cpu().numpy().any(), (n, nrhs)
        err = relerr(dX.cpu().numpy(), np.linalg.solve(A, B))
        worst = max(worst, err)
        assert err < 1e-10, (n, nrhs, err)
    print(f"large orders: worst relative error {worst:.2e} over {len(sizes)} orders")


def test_zpotrs_size_sweep(T):
    """Every order 1..70 and a few around the 16 / 32 block boundaries further up, with 1..40 right-hand
    sides: exercises the 16-wide last block column, single-tile groups, 2+2 / 3 groupings and the
    k-sliced / unsliced forms of the back substitution."""
    from hydra_pspec_amd import hpx
    rng = np.random.default_rng(99)
    sizes = list(range(1, 71)) + [95, 96, 97, 111, 112, 113, 127, 128, 129, 160, 161, 255, 256, 257]
    worst = 0.0
    for i, n in enumerate(sizes):
        nrhs = [1, 2, 15, 16, 17, 31, 32, 33, 40][i % 9]
        nb = 1 + i % 3
        A = _hpd(rng, nb, n, cond=1e3)
        A = 0.5 * (A + np.conj(np.swapaxes(A, 1, 2)))
        B = rng.standard_normal((nb, n, nrhs)) + 1j * rng.standard_normal((nb, n, nrhs))
        dA, dB = _dev(T, A, T.complex128), _dev(T, B, T.complex128)
        dX = T.zeros_like(dB)
        info = T.zeros(nb, dtype=T.int32, device="cuda")
        hpx.check(hpx.lib().hpx_zpotrs_batched(nb, n, nrhs, hpx.ptr(dA), hpx.ptr(dB), hpx.ptr(dX),
                                               hpx.ptr(info), None))
        assert not info.cpu().numpy().any(), (n, nrhs)
        err = relerr(dX.cpu().numpy(), np.linalg.solve(A, B))
        worst = max(worst, err)
        assert err < 1e-10, (n, nrhs, err)
    print(f"size sweep: worst relative error {worst:.2e} over {len(sizes)} orders")


@pytest.mark.parametrize("n,flagged", [(48, False), (100, True), (512, True)])
def test_sqrtm_hpd_on_device(T, n, flagged):
    """hpx_sqrtm_hpd_batched (Newton-Schulz on the batched MFMA product) and the masked root built on it
    (pspec.sqrtm_masked_device) against scipy.linalg.sqrtm of Ninv diag(w) -- the reference's own call, pspec.py:361-362."""
    import scipy.linalg
    from hydra_pspec_amd import hpx, pspec
    rng = np.random.default_rng(n)
    nb = 3
    q = rng.standard_normal((nb, n, n)) + 1j * rng.standard_normal((nb, n, n))
    A = q @ np.conj(np.swapaxes(q, 1, 2)) / n + 0.5 * np.eye(n)
    w = np.ones((nb, n), bool)
    if flagged:
        w = rng.uniform(size=(nb, n)) > 0.15
        w[0] = True                                  # (one system without flags in the same batch)
    got = pspec.sqrtm_masked_device(T, A, w, T.device("cuda", T.cuda.current_device())).cpu().numpy()
    for b in range(nb):
        Ni = A[b] * w[b][None, :]
        assert np.abs(got[b] @ got[b] - Ni).max() < 1e-10 * np.abs(Ni).max()
        if n <= 100:
            assert np.abs(got[b] - scipy.linalg.sqrtm(Ni)).max() < 1e-9 * np.abs(Ni).max()
    if not flagged and n % 16 == 0:                  # the C-ABI entry point itself: root and inverse root
        dA = T.from_numpy(np.ascontiguousarray(A)).cuda()
        sq, isq = T.empty_like(dA), T.empty_like(dA)
        hpx.check(hpx.lib().hpx_sqrtm_hpd_batched(nb, n, hpx.ptr(dA), hpx.ptr(sq), hpx.ptr(isq), 1e-7, 60, None, None))
        sq, isq = sq.cpu().numpy(), isq.cpu().numpy()
        assert np.abs(sq @ sq - A).max() < 1e-11 * np.abs(A).max()
        assert np.abs(sq @ isq - np.eye(n)).max() < 1e-11


@pytest.mark.parametrize("nb", [1, 3])
def test_sqrtm_indefinite_input_is_refused(T, nb):
    """A matrix that is not positive definite makes Newton-Schulz overflow to inf, then NaN: the library must say so
    (HPX_EINVAL), also when it is the ONLY matrix of the call (a NaN residual used to be dropped by the max and read as
    converged), and pspec.sqrtm_masked_device must raise the reference-side error instead of handing NaN roots to the
    chain."""
    from hydra_pspec_amd import hpx, pspec
    n = 48
    rng = np.random.default_rng(5)
    q = rng.standard_normal((nb, n, n)) + 1j * rng.standard_normal((nb, n, n))
    A = q @ np.conj(np.swapaxes(q, 1, 2)) / n + 0.5 * np.eye(n)
    lam, V = np.linalg.eigh(A[-1])
    lam[0] = -0.7 * lam[-1]                           # one clearly negative eigenvalue
    A[-1] = (V * lam) @ V.conj().T
    dA = T.from_numpy(np.ascontiguousarray(A)).cuda()
    sq, isq = T.empty_like(dA), T.empty_like(dA)
    rc = hpx.lib().hpx_sqrtm_hpd_batched(nb, n, hpx.ptr(dA), hpx.ptr(sq), hpx.ptr(isq), 1e-7, 60, None, None)
    assert rc == hpx.HPX_EINVAL, rc
    with pytest.raises(FloatingPointError, match="not positive definite"):
        pspec.sqrtm_masked_device(T, A, np.ones((nb, n), bool), T.device("cuda", T.cuda.current_device()))
