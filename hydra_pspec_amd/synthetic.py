"""Synthetic per-baseline inputs for the Gibbs path (SURVEY 8d recipe).

Used by bench.py, the tests and tests/golden/make_golden.py.  Deterministic in
``(N, T, M, flag_frac, baseline index)``: baseline ``k`` draws everything from
``numpy.random.default_rng(1000 + k)``.

Model (all baselines share the spectrum shape, the DPSS foreground basis, the
prior window and the chain seed, as in the reference's own scaling study which
replicates one baseline -- scripts/scaling_tests/set_up_scaling_data.py:19-34):

* EoR delay spectrum ``p_j = 16 ** (-|j - N//2| / (N/2))``; ``S = F^H diag(p/N^2) F``;
  ``e_t = F^H (sqrt(p) * z_t) / N`` so that ``E|F e|^2 = p``.
* noise: complex white, ``E|n|^2 = sigma_n^2``, ``sigma_n = sigma_e/10``,
  ``sigma_e^2 = mean(p)/N`` (= diag S); ``Ninv = I / sigma_n^2``.
* foregrounds: ``fgmodes = dpss(N, NW=M/2, Kmax=M, sym=False).T``; amplitudes
  ``a_tm = 2000 sigma_e sqrt(N) z / (1+m)^2``.
* flags: ``round(flag_frac*N)`` channels, chosen without replacement, set
  False for all times (True = use the channel).
* prior: 7 bins around the centre, ``hi=2.0`` (row 0), ``lo=0.1`` (row 1).
"""
import numpy as np
from scipy.signal.windows import dpss

from .utils import fourier_operator

CHAIN_SEED = 7123689  # test_data/config.yaml:3


def _cnormal(rng, shape):
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) / np.sqrt(2.0)


def true_pspec(N):
    j = np.arange(N)
    return 16.0 ** (-np.abs(j - N // 2) / (N / 2))


def default_prior(N, n_bins=3, lo=0.1, hi=2.0):
    """(2,N) prior box, rows [hi, lo] (run-hydra-pspec.py:509-517)."""
    pr = np.zeros((2, N))
    sl = slice(N // 2 - n_bins, N // 2 + n_bins + 1)
    pr[0, sl] = hi
    pr[1, sl] = lo
    return pr


def make_baselines(N, T, M, k0=0, nbl=1, flag_frac=0.0, prior=True, dense=True):
    """Return a dict of stacked inputs for baselines ``k0 .. k0+nbl-1``.

    keys: vis (nbl,T,N) c128, flags (nbl,N) bool, ps0 (N,) f64 [= true P(k), the
    initial bandpowers], S_initial (N,N) c128 (only if ``dense``), fgmodes (N,M)
    f64, ninv_diag (nbl,N) f64, Ninv (N,N) f64 (only if ``dense``; same for all
    baselines), ps_prior (2,N), seed."""
    p = true_pspec(N)
    sig_e = np.sqrt(p.mean() / N)
    sig_n = sig_e / 10.0
    F = np.ascontiguousarray(dpss(N, NW=M / 2.0, Kmax=M, sym=False).T)
    fscale = 2000.0 * sig_e * np.sqrt(N) / (1.0 + np.arange(M)) ** 2
    vis = np.empty((nbl, T, N), dtype=complex)
    flags = np.ones((nbl, N), dtype=bool)
    sqp = np.sqrt(p)
    for i in range(nbl):
        rng = np.random.default_rng(1000 + k0 + i)
        # e = F^H (sqrt(p) z) / N, evaluated as the centred inverse FFT
        e = np.fft.fftshift(np.fft.ifft(np.fft.ifftshift(sqp[None, :] * _cnormal(rng, (T, N)), axes=1),
                                        axis=1), axes=1)
        a = _cnormal(rng, (T, M)) * fscale[None, :]
        noise = sig_n * _cnormal(rng, (T, N))
        vis[i] = e + a @ F.T + noise
        nfl = int(round(flag_frac * N))
        if nfl:
            flags[i, rng.choice(N, size=nfl, replace=False)] = False
    out = dict(vis=vis, flags=flags, ps0=p, fgmodes=F,
               ninv_diag=np.full((nbl, N), 1.0 / sig_n ** 2),
               ps_prior=default_prior(N) if prior else np.zeros((2, N)),
               seed=CHAIN_SEED, sigma_n=sig_n)
    if dense:
        fop = fourier_operator(N)
        out["S_initial"] = fop.conj().T @ np.diag(p / N ** 2) @ fop
        out["Ninv"] = np.eye(N) / sig_n ** 2
    return out
