"""Foreground modes from the data: the batched Jacobi eigensolver behind
``fgmodes.cov_eig_modes`` against numpy's eigendecomposition of ``np.cov(bl_data.T)``
(reference scripts/calc-vis-cov-matrices.py:235-249)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fg_like(rng, nbl, T, N, nstrong=5, base=10.0):
    """Smooth-spectrum 'foregrounds' with a steep eigenvalue spectrum plus a little noise."""
    nu = np.linspace(-1, 1, N)
    basis = np.stack([np.cos(np.pi * k * nu / 2 + 0.3 * k) * np.exp(0.2j * k * nu) for k in range(nstrong)], axis=1)
    amps = (rng.standard_normal((nbl, T, nstrong)) + 1j * rng.standard_normal((nbl, T, nstrong))) \
        * (base ** -np.arange(nstrong))[None, None, :] * 100
    noise = 1e-6 * (rng.standard_normal((nbl, T, N)) + 1j * rng.standard_normal((nbl, T, N)))
    return amps @ basis.T + 3.0 + noise                 # + a constant that np.cov removes


@pytest.mark.parametrize("nbl,T,N,nm", [(3, 32, 96, 4), (2, 40, 24, 4), (2, 31, 64, 3), (1, 60, 25, 5),
                                        (2, 203, 120, 4), (2, 32, 512, 12), (1, 300, 320, 4), (1, 330, 300, 3)])
def test_cov_eig_modes_vs_numpy(nbl, T, N, nm):
    from hydra_pspec_amd import fgmodes
    rng = np.random.default_rng(100 * T + N)
    vis = _fg_like(rng, nbl, T, N) if nm <= 5 else _fg_like(rng, nbl, T, N, nstrong=nm + 2, base=2.0)
    modes, evals = fgmodes.cov_eig_modes(vis, nm, return_evals=True)
    assert modes.shape == (nbl, N, nm) and evals.shape == (nbl, nm)
    for b in range(nbl):
        C = np.cov(vis[b].T)
        lam, U = np.linalg.eigh(C)
        lam, U = lam[::-1][:nm], U[:, ::-1][:, :nm]
        assert np.allclose(evals[b], lam, rtol=1e-9)
        # eigenvectors agree up to a phase; ours are unit norm with the largest component real > 0
        assert np.allclose(np.linalg.norm(modes[b], axis=0), 1.0, rtol=1e-12)
        for m in range(nm):
            k = np.argmax(np.abs(modes[b][:, m]))
            assert abs(modes[b][k, m].imag) < 1e-12 and modes[b][k, m].real > 0
            overlap = abs(np.vdot(U[:, m], modes[b][:, m]))
            assert overlap > 1 - 1e-9
        # and they are eigenvectors of the covariance
        assert np.max(np.abs(C @ modes[b] - modes[b] * evals[b][None, :])) < 1e-9 * lam[0]


def test_cov_eig_modes_single_baseline_and_errors():
    from hydra_pspec_amd import fgmodes
    rng = np.random.default_rng(5)
    vis = _fg_like(rng, 2, 16, 48)
    both = fgmodes.cov_eig_modes(vis, 3)
    one = fgmodes.cov_eig_modes(vis[1], 3)
    assert one.shape == (48, 3) and np.array_equal(one, both[1])
    with pytest.raises(ValueError):
        fgmodes.cov_eig_modes(vis, 16)                   # rank of the covariance is Ntimes - 1
    with pytest.raises(NotImplementedError):
        fgmodes.cov_eig_modes(np.zeros((1, 1030, 1040), complex), 2)


def test_modes_drive_the_sampler():
    """End to end: modes estimated from a foreground-only cube are a usable `fgmodes` input."""
    from hydra_pspec_amd import fgmodes, pspec, synthetic
    d = synthetic.make_baselines(64, 16, 6, k0=3, nbl=2, dense=False)
    rng = np.random.default_rng(8)
    fg_only = (rng.standard_normal((2, 16, 6)) + 1j * rng.standard_normal((2, 16, 6))) @ d["fgmodes"].T * 50
    F = fgmodes.cov_eig_modes(fg_only, 6)                # (nbl, N, 6): per-baseline modes
    # the estimated modes span the true ones
    for b in range(2):
        P = F[b] @ F[b].conj().T
        assert np.max(np.abs(P @ d["fgmodes"] - d["fgmodes"])) < 1e-8 * np.abs(d["fgmodes"]).max()
    out = pspec.gibbs_sample_with_fg_batched(d["vis"], d["flags"], F, d["ninv_diag"], d["ps_prior"],
                                             ps_initial=d["ps0"], Niter=3, seed=2)
    assert np.isfinite(out["signal_ps"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("n0,kind", [(512, "cov"), (1024, "cov"), (330, "gram"), (144, "hpd")])
def test_zheev_psd_batched_vs_numpy(n0, kind):
    """hpx_zheev_psd_batched (blocked one-sided Jacobi on the Cholesky factor, csrc/hpx_eigh.hip) against
    numpy.linalg.eigh at the orders the covariance path meets (512 and 1024): eigenvalues to 1e-9 of the
    largest, residual and orthogonality of the full eigenvector set, overlap of the 12 leading eigenvectors -- for a
    foreground-dominated covariance (T > n samples), the exactly singular Gram matrix of a centred cube and a generic
    positive definite matrix."""
    import ctypes
    import torch
    from hydra_pspec_amd import hpx
    rng = np.random.default_rng(n0)
    nb = 2
    if kind == "cov":
        T = n0 + 24
        modes = rng.standard_normal((n0, 12)) + 1j * rng.standard_normal((n0, 12))
        amp = (rng.standard_normal((nb, T, 12)) + 1j * rng.standard_normal((nb, T, 12))) * np.logspace(3, 0.5, 12)
        x = amp @ modes.T + (rng.standard_normal((nb, T, n0)) + 1j * rng.standard_normal((nb, T, n0)))
        A = np.stack([np.cov(x[b].T) for b in range(nb)])
    elif kind == "gram":
        x = rng.standard_normal((nb, n0, n0 + 50)) + 1j * rng.standard_normal((nb, n0, n0 + 50))
        x = x - x.mean(axis=1, keepdims=True)
        A = np.stack([x[b].conj() @ x[b].T / (n0 - 1) for b in range(nb)])
    else:
        q = rng.standard_normal((nb, n0, n0)) + 1j * rng.standard_normal((nb, n0, n0))
        A = q @ np.conj(np.swapaxes(q, 1, 2)) / n0 + np.eye(n0)
    A = 0.5 * (A + np.conj(np.swapaxes(A, 1, 2)))
    n = hpx.lib().hpx_zheev_psd_order(n0)
    dA = torch.from_numpy(np.ascontiguousarray(A)).cuda()
    w = torch.empty((nb, n), dtype=torch.float64, device="cuda")
    v = torch.empty((nb, n0, n), dtype=torch.complex128, device="cuda")
    sw = ctypes.c_int(0)
    hpx.check(hpx.lib().hpx_zheev_psd_batched(nb, n0, hpx.ptr(dA), hpx.ptr(w), hpx.ptr(v), ctypes.byref(sw), None))
    w, v = w.cpu().numpy(), v.cpu().numpy()
    lam, U = np.linalg.eigh(A)
    assert 0 < sw.value < 30
    for b in range(nb):
        order = np.argsort(-w[b])[:n0]
        wb, vb = w[b][order], v[b][:, order]
        lr, Ur = lam[b][::-1], U[b][:, ::-1]
        assert np.max(np.abs(wb - lr)) < 1e-9 * lr[0]
        assert np.linalg.norm(A[b] @ vb - vb * wb[None, :]) < 1e-9 * np.linalg.norm(A[b])
        keep = wb > 1e-9 * lr[0]                       # (a null vector competes with the zero padding's)
        assert np.abs(vb[:, keep].conj().T @ vb[:, keep] - np.eye(int(keep.sum()))).max() < 1e-9
        if kind != "hpd":                             # well separated leading eigenvalues
            assert np.max(1 - np.abs(np.sum(np.conj(Ur[:, :12]) * vb[:, :12], axis=0))) < 1e-9


@pytest.mark.gpu
def test_cov_eig_modes_more_times_than_channels():
    """Ntimes > Nfreqs: the reference diagonalises the Nfreqs x Nfreqs covariance itself
    (scripts/calc-vis-cov-matrices.py:239-247); here at Nfreqs = 256 through the blocked solver."""
    from hydra_pspec_amd import fgmodes
    rng = np.random.default_rng(8)
    nbl, T, N, nm = 3, 300, 256, 10
    modes = rng.standard_normal((N, nm)) + 1j * rng.standard_normal((N, nm))
    amp = (rng.standard_normal((nbl, T, nm)) + 1j * rng.standard_normal((nbl, T, nm))) * np.logspace(2.5, 0.5, nm)
    vis = amp @ modes.T + 0.1 * (rng.standard_normal((nbl, T, N)) + 1j * rng.standard_normal((nbl, T, N)))
    got, evals = fgmodes.cov_eig_modes(vis, nm, return_evals=True)
    for b in range(nbl):
        lam, U = np.linalg.eigh(np.cov(vis[b].T))
        lam, U = lam[::-1][:nm], U[:, ::-1][:, :nm]
        assert np.max(np.abs(evals[b] / lam - 1)) < 1e-9
        assert np.max(1 - np.abs(np.sum(U.conj() * got[b], axis=0))) < 1e-9


@pytest.mark.gpu
def test_zheev_psd_refuses_non_finite_input():
    """A NaN in the input reaches the Gram blocks: the convergence measure must not read it as "orthogonal already"
    (fmax and the host's max both drop NaN): HPX_EINVAL, sweeps -1."""
    import ctypes
    import torch
    from hydra_pspec_amd import hpx
    n0, nb = 144, 2
    rng = np.random.default_rng(1)
    q = rng.standard_normal((nb, n0, n0)) + 1j * rng.standard_normal((nb, n0, n0))
    A = q @ np.conj(np.swapaxes(q, 1, 2)) / n0 + np.eye(n0)
    A[1, 5, 5] = np.nan
    n = hpx.lib().hpx_zheev_psd_order(n0)
    dA = torch.from_numpy(np.ascontiguousarray(A)).cuda()
    w = torch.empty((nb, n), dtype=torch.float64, device="cuda")
    v = torch.empty((nb, n0, n), dtype=torch.complex128, device="cuda")
    sw = ctypes.c_int(0)
    rc = hpx.lib().hpx_zheev_psd_batched(nb, n0, hpx.ptr(dA), hpx.ptr(w), hpx.ptr(v), ctypes.byref(sw), None)
    assert rc in (hpx.HPX_EINVAL, hpx.HPX_ENOTPD), rc     # (the Cholesky in front may refuse it first)
