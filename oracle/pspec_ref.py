"""Operation-by-operation numpy/scipy restatement of the reference Gibbs path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Every function names the
reference lines (under /root/reference) whose arithmetic it follows.  The
restatement keeps the reference's expensive choices on purpose (two ``sqrtm``,
``pinv``, the non-Hermitian block system, scipy CG at rtol=1e-8, the
per-channel draw loop) so that it (a) reproduces the reference's numbers,
including its solver noise, and (b) costs what the reference costs when it is
timed as ``cpu_baseline`` (kind "port").  The only structural change is that
the per-time solves run in a plain loop instead of a forked
``multiprocess.Pool``: the reference reseeds inside each forked task, so its
results do not depend on the pool, and the parent RNG stream is untouched by
the children -- emulated here by saving/restoring the global RNG state.

``solver="direct"`` swaps CG for ``numpy.linalg.solve`` (the "reference with an
exact solve" control chain of SURVEY 8c/T2); it is not the reference's
behaviour and is never used for golden comparison.
"""
import numpy as np
import scipy.linalg
import scipy.sparse.linalg
from scipy.interpolate import interp1d
from scipy.stats import invgamma

GCR_SEED0 = 912983  # pspec.py:153 (multiprocess_seed default, never overridden)


def fourier_operator(n):
    """Centred DFT matrix, utils.py:15-41 (same operation order => same bits)."""
    ix = (np.arange(n) - n // 2).reshape(1, -1)
    ik = (np.arange(n) - n // 2).reshape(-1, 1)
    return np.exp(-2 * np.pi * 1j * (ik * ix / n))


def covariance_from_pspec(ps, fourier_op):
    """F^H diag(ps) F, pspec.py:313-322."""
    n = ps.size
    dg = np.zeros((n, n), dtype=complex)
    dg[np.diag_indices(n)] = ps
    return fourier_op.T.conj() @ dg @ fourier_op


def inversion_sample_invgamma(alpha, beta, prior_min, prior_max, ngrid=1000):
    """Truncated inverse-gamma draw by CDF inversion, pspec.py:11-64.

    Consumes exactly one ``np.random.uniform()`` from the global stream."""
    if prior_min <= 0:
        raise ValueError("prior_min must be greater than zero")
    if prior_max <= 0:
        raise ValueError("prior_max must be greater than zero")
    if not np.isfinite(prior_max):
        raise ValueError("prior_max must be finite")
    if prior_max <= prior_min:
        raise ValueError("prior_max must be greater than prior_min")
    x = np.logspace(np.log10(prior_min), np.log10(prior_max), ngrid)
    cdf = invgamma.cdf(x, a=alpha, loc=0, scale=beta)
    cdf -= cdf.min()
    cdf /= cdf.max()
    cdf_u, idx_u = np.unique(cdf, return_index=True)
    u = np.random.uniform()
    return interp1d(cdf_u, x[idx_u], kind="linear")(u)


def sample_S(s=None, sk=None, prior=None):
    """Bandpower draw p(S|s), pspec.py:67-127.

    One uniform per channel from the global stream, in channel order
    (``invgamma.rvs`` = ppf(U) in scipy>=1.15: SURVEY 8a row P5)."""
    if s is None and sk is None:
        raise ValueError("Must pass in s (real space) or sk (Fourier space) vector.")
    if sk is None:
        sk = np.fft.fftshift(np.fft.fftn(np.fft.ifftshift(s, axes=(1,)), axes=(1,)), axes=(1,))
    nobs, nfreq = sk.shape
    if prior is None:
        prior = np.zeros((2, nfreq))
    beta = np.sum(sk * sk.conj(), axis=0).real
    alpha = nobs - 1.0
    x = np.zeros(nfreq)
    for i in range(nfreq):
        if np.any(prior[:, i] > 0):
            x[i] = inversion_sample_invgamma(alpha + 1, beta[i], prior[1, i], prior[0, i])
        else:
            x[i] = invgamma.rvs(a=alpha) * beta[i]
    return x


def sprior(signals, bins, factor):
    """Data-derived prior box, pspec.py:130-148."""
    nobs, nfreq = signals.shape
    sk = np.fft.fft(signals, axis=-1)
    ds = np.sum(sk * sk.conj(), axis=0).real
    prior = np.zeros((2, nfreq))
    prior[0] = ds * factor
    prior[1] = ds / factor
    prior[0, bins + 1:-bins] = 0
    prior[1, bins + 1:-bins] = 0
    return prior / (nobs / 2 - 1)


def build_matrices(nparams, flags, signal_S, Ninv, fgmodes):
    """Operators of the GCR system, pspec.py:325-374.

    Returns ``[ (4,N,N)=[Sh,S,Ni,Nih], (2,n,n)=[A, pinv(A)] ]``."""
    n = signal_S.shape[0]
    ops = np.zeros((4, n, n), dtype=complex)
    ops[0] = scipy.linalg.sqrtm(signal_S)
    ops[1] = signal_S.copy()
    ops[2] = flags.T * Ninv * flags          # column mask only (reference FIXME, :361)
    ops[3] = scipy.linalg.sqrtm(ops[2])
    S, Ni = ops[1], ops[2]
    A = np.zeros((nparams, nparams), dtype=complex)
    A[:n, :n] = np.eye(n) + S @ Ni
    A[:n, n:] = S @ Ni @ fgmodes
    A[n:, :n] = fgmodes.T.conj() @ Ni
    A[n:, n:] = fgmodes.T.conj() @ Ni @ fgmodes
    sysm = np.zeros((2, nparams, nparams), dtype=complex)
    sysm[0] = A
    sysm[1] = np.linalg.pinv(A)
    return [ops, sysm]


def gcr_fgmodes_1d(idx, vis, w, matrices, fgmodes, f0=None, map_estimate=False,
                   verbose=False, multiprocess_seed=GCR_SEED0, solver="cg"):
    """One time sample of the constrained-realisation solve, pspec.py:151-235.

    Reseeds the GLOBAL legacy RNG with ``multiprocess_seed + idx`` (:196-197)
    and draws omi, omj, omk, oml in that order (:215-216)."""
    np.random.seed(multiprocess_seed + idx)
    nfreq, nmodes = fgmodes.shape
    d = vis.reshape((1, max(nfreq, len(vis.T))))
    Sh, S, Ni, Nih = matrices[0][0], matrices[0][1], matrices[0][2], matrices[0][3]
    A, Ai = matrices[1][0], matrices[1][1]
    if map_estimate:
        oma = np.zeros((nfreq, 1), dtype=complex)
        omb = np.zeros((nfreq, 1), dtype=complex)
    else:
        omi, omj = np.random.randn(nfreq, 1), np.random.randn(nfreq, 1)
        omk, oml = np.random.randn(nfreq, 1), np.random.randn(nfreq, 1)
        oma, omb = (omi + 1.0j * omj) / 2 ** 0.5, (omk + 1.0j * oml) / 2 ** 0.5
    b = np.zeros((nfreq + nmodes, 1), dtype=complex)
    b[:nfreq] = S @ Ni @ (w * d).T + Sh @ oma + S @ Nih @ omb
    b[nfreq:] = fgmodes.T.conj() @ (Ni @ (w * d).T + Nih @ omb)
    if solver == "cg":
        x0 = None
        if f0 is not None:
            x0 = np.concatenate((np.zeros(nfreq, dtype=complex), f0))
        x, info = scipy.sparse.linalg.cg(A, b, maxiter=int(1e5), rtol=1e-8, atol=1e-6,
                                         x0=x0, M=Ai)
    else:
        x, info = np.linalg.solve(A, b[:, 0]), 0
    resid = np.abs(A @ x - b[:, 0]).mean() if verbose else None
    return x, resid, info


def gcr_fgmodes(vis, w, matrices, fgmodes, f0=None, nproc=1, map_estimate=False,
                verbose=False, solver="cg"):
    """All time samples, pspec.py:238-310 (pool replaced by a loop; the parent
    RNG state is preserved exactly as a forked child would leave it)."""
    keep = np.random.get_state()
    rows = []
    try:
        for t in range(vis.shape[0]):
            x, _, _ = gcr_fgmodes_1d(t, vis[t], w, matrices, fgmodes, f0=f0,
                                     map_estimate=map_estimate, verbose=verbose,
                                     solver=solver)
            rows.append(x)
    finally:
        np.random.set_state(keep)
    return np.array(rows).reshape((vis.shape[0], -1))


def gibbs_step_fgmodes(vis, flags, signal_S, fgmodes, Ninv, ps_prior=None, f0=None,
                       nproc=1, map_estimate=False, verbose=False, solver="cg"):
    """One Gibbs iteration, pspec.py:377-490."""
    nfreq = vis.shape[1]
    nmodes = fgmodes.shape[1]
    assert flags.shape == (nfreq,), "`flags` array must have shape (Nfreqs,)"
    fop = fourier_operator(nfreq)
    mats = build_matrices(nfreq + nmodes, flags, signal_S, Ninv, fgmodes)
    cr = gcr_fgmodes(vis, flags, mats, fgmodes, f0=f0, nproc=nproc,
                     map_estimate=map_estimate, verbose=verbose, solver=solver)
    signal_cr = cr[:, :-nmodes]
    fg_amps = cr[:, -nmodes:]
    model = signal_cr + fg_amps @ fgmodes.T
    chisq = np.abs(vis - model) ** 2 * Ninv.diagonal()[None, :]
    ps_sample = sample_S(s=signal_cr, prior=ps_prior)
    S_sample = covariance_from_pspec(ps_sample / nfreq ** 2, fop)
    Sinv = np.linalg.inv(S_sample)
    r = (vis - model)[:, flags]
    sf = signal_cr[:, flags]
    ln_post = np.sum(np.diagonal(-(r.conj() @ Ninv[flags][:, flags] @ r.T)
                                 - (sf.conj() @ Sinv[flags][:, flags] @ sf.T))).real
    return signal_cr, S_sample, ps_sample, fg_amps, chisq, ln_post


def gibbs_sample_with_fg(vis, flags, S_initial, fgmodes, Ninv, ps_prior, Niter=100,
                         seed=None, verbose=False, nproc=1, write_Niter=100,
                         out_dir=None, map_estimate=False, solver="cg", ps_forced=None):
    """Chain driver, pspec.py:493-658 (file output omitted: out_dir ignored).

    ``ps_forced`` (Niter,N), optional: teacher forcing -- before iteration i>0
    the covariance is rebuilt from ``ps_forced[i-1]`` instead of the chain's own
    draw (used to build per-step parity tests; with the reference's own chain as
    ``ps_forced`` this is the identity)."""
    if map_estimate:
        Niter = 1
    else:
        np.random.seed(seed)
    T, N = vis.shape
    M = fgmodes.shape[1]
    assert flags.shape == (N,), "`flags` array must have shape (Nfreqs,)"
    assert fgmodes.shape[0] == N, "fgmodes must have shape (Nfreqs, Nmodes)"
    signal_cr = np.zeros((Niter, T, N), dtype=complex)
    signal_ps = np.zeros((Niter, N))
    fg_amps = np.zeros((Niter, T, M), dtype=complex)
    chisq = np.zeros((Niter, T, N))
    ln_post = np.zeros(Niter)
    signal_S = S_initial.copy()
    fop = fourier_operator(N) if ps_forced is not None else None
    for i in range(Niter):
        if ps_forced is not None and i > 0:
            signal_S = covariance_from_pspec(ps_forced[i - 1] / N ** 2, fop)
        signal_cr[i], signal_S, signal_ps[i], fg_amps[i], chisq[i], ln_post[i] = \
            gibbs_step_fgmodes(vis * flags, flags, signal_S, fgmodes, Ninv, ps_prior,
                               f0=None, nproc=nproc, map_estimate=map_estimate,
                               verbose=False, solver=solver)
    return signal_cr, signal_S, signal_ps, fg_amps, chisq, ln_post, 0.0


# ---- time-dependent flags / noise ------------------------------------------------------
# The reference documents this mode (Ninv of shape (Ntimes, Nfreqs, Nfreqs): pspec.py:337-340, :398-401)
# but does not implement it (FIXMEs :361, :450-451); its driver reduces per-time flags to an any-time
# mask (run-hydra-pspec.py:524-541).  What follows is the reference's own per-time solve
# (gcr_fgmodes_1d, :151-235) given, for time t, the operators build_matrices (:325-374) makes from
# flags_t[t] and Ninv_t[t] -- the natural completion of the FIXMEs -- with chi^2 and the two
# ln-posterior terms (:452, :472-485) taken per time with that time's flags and noise.  With
# flags_t[t] = flags and Ninv_t[t] = Ninv for every t it is gibbs_step_fgmodes above, operation for
# operation (tests/test_oracle_golden.py pins that against the reference's golden steps).
def gibbs_step_fgmodes_pertime(vis, flags_t, signal_S, fgmodes, Ninv_t, ps_prior=None, map_estimate=False,
                               solver="direct"):
    """``vis`` (T,N) already multiplied by flags_t; ``flags_t`` (T,N) bool; ``Ninv_t`` (T,N) diagonals or
    (T,N,N) matrices (the shape the reference's docstrings name, pspec.py:337-340)."""
    T, nfreq = vis.shape
    nmodes = fgmodes.shape[1]
    assert flags_t.shape == (T, nfreq) and Ninv_t.shape in ((T, nfreq), (T, nfreq, nfreq))
    full = Ninv_t.ndim == 3
    fop = fourier_operator(nfreq)
    keep = np.random.get_state()
    rows = []
    try:
        for t in range(T):
            mats = build_matrices(nfreq + nmodes, flags_t[t], signal_S,
                                  Ninv_t[t] if full else np.diag(Ninv_t[t]).astype(complex), fgmodes)
            x, _, _ = gcr_fgmodes_1d(t, vis[t], flags_t[t], mats, fgmodes, map_estimate=map_estimate, solver=solver)
            rows.append(x)
    finally:
        np.random.set_state(keep)
    cr = np.array(rows).reshape((T, -1))
    signal_cr = cr[:, :-nmodes]
    fg_amps = cr[:, -nmodes:]
    model = signal_cr + fg_amps @ fgmodes.T
    chisq = np.abs(vis - model) ** 2 * (np.diagonal(Ninv_t, axis1=1, axis2=2).real if full else Ninv_t)   # :452
    ps_sample = sample_S(s=signal_cr, prior=ps_prior)
    S_sample = covariance_from_pspec(ps_sample / nfreq ** 2, fop)
    Sinv = np.linalg.inv(S_sample)
    ln_post = 0.0
    for t in range(T):
        f = flags_t[t]
        r = (vis[t] - model[t])[f]
        sf = signal_cr[t][f]
        nr = Ninv_t[t][f][:, f] @ r if full else Ninv_t[t][f] * r                       # :472-477 per time
        ln_post += (-(r.conj() @ nr) - (sf.conj() @ Sinv[f][:, f] @ sf)).real
    return signal_cr, S_sample, ps_sample, fg_amps, chisq, ln_post


def gibbs_sample_with_fg_pertime(vis, flags_t, S_initial, fgmodes, Ninv_t, ps_prior, Niter=100, seed=None,
                                 solver="direct"):
    np.random.seed(seed)
    T, N = vis.shape
    M = fgmodes.shape[1]
    signal_cr = np.zeros((Niter, T, N), dtype=complex)
    signal_ps = np.zeros((Niter, N))
    fg_amps = np.zeros((Niter, T, M), dtype=complex)
    chisq = np.zeros((Niter, T, N))
    ln_post = np.zeros(Niter)
    signal_S = S_initial.copy()
    for i in range(Niter):
        signal_cr[i], signal_S, signal_ps[i], fg_amps[i], chisq[i], ln_post[i] = \
            gibbs_step_fgmodes_pertime(vis * flags_t, flags_t, signal_S, fgmodes, Ninv_t, ps_prior, solver=solver)
    return signal_cr, signal_S, signal_ps, fg_amps, chisq, ln_post, 0.0
