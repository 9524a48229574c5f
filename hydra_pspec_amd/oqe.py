"""Optimal-quadratic-estimator helpers (mirror of ``hydra_pspec.oqe``, reference
hydra_pspec/oqe.py).  Same names and return values; the shipped reference file
cannot run as is (it uses ``os``, ``sp`` and ``time`` without importing them and
caches ``Q`` matrices under ``Qs/``), so "reference behaviour" means the behaviour
with those names provided (tests/golden/make_golden.py).

The two expensive pieces run on the GPU: the Fisher matrices ``F`` / ``Ft``
(O(s^5) / O(s^4) trace loops in the reference, here two s x s x s contractions:
``X = M R M^H``, ``Wm = conj(M) R M^T``, ``F_ab = conj(Wm_ba) X_ab / 2``,
``Ft_ab = |X_ab|^2 / 2`` -- valid for any ``R``) and the cross-estimator ``q_h``
(``q_t = conj(FFT(R x1))_t FFT(R x2)_t / 2``), both as dense MFMA products with the DFT matrix;
the normalisations (``M_Finv``, ``M_Fhalf`` on the batched Cholesky solver, ``M_opt``) and the
noise terms (``bias``, ``Sig_QEN``, ``Sig_QESN``: diagonals of ``M (R C R') M^H``) run on the
device as well.
"""
import numpy as np
import scipy.linalg

from . import hpx


def m(tau, s):                      # oqe.py:7-10
    y = np.zeros(s)
    y[tau] = 1
    return np.fft.fft(y)


def Q(tau, s):                      # oqe.py:13-20 (computed, not cached on disk)
    v = m(tau, s)
    return np.outer(v.conj(), v)


def _dev():
    torch = hpx.require_gpu()
    return torch, torch.device("cuda", torch.cuda.current_device())


def _c128(torch, dev, x):
    return hpx.to_dev(torch, np.ascontiguousarray(np.asarray(x, dtype=complex)), torch.complex128, dev)


def _fisher(R, variant):
    torch, dev = _dev()
    R = np.asarray(R, dtype=complex)
    batched = R.ndim == 3
    Rb = R if batched else R[None]
    nb, s, _ = Rb.shape
    d_R = _c128(torch, dev, Rb)
    d_F = torch.zeros_like(d_R)
    hpx.check(hpx.lib().hpx_oqe_fisher(nb, s, hpx.ptr(d_R), hpx.ptr(d_F), variant, None, 0,
                                       hpx.stream_ptr(torch)), "hpx_oqe_fisher")
    out = d_F.cpu().numpy()
    return out if batched else out[0]


def F(s, R):
    """Fisher matrix ``F_ab = tr(R* Q_a R Q_b) / 2`` (oqe.py:43-50); ``R`` (s,s) or a batch (nb,s,s)."""
    assert np.shape(R)[-1] == s
    return _fisher(R, 0)


def Ft(s, R):
    """``Ft_ab = tr(R^H Q_a R Q_b) / 2`` (oqe.py:53-66)."""
    assert np.shape(R)[-1] == s
    return _fisher(R, 1)


def _device_inverse(torch, dev, d_A):
    """inv(A) for a Hermitian positive-definite (1,s,s) device matrix through the batched Cholesky
    solver (identity right-hand side); None if a pivot is not positive."""
    s = d_A.shape[-1]
    d_eye = torch.eye(s, dtype=torch.complex128, device=dev)[None].contiguous()
    d_inv = torch.empty_like(d_A)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    hpx.check(hpx.lib().hpx_zpotrs_batched(1, s, s, hpx.ptr(d_A), hpx.ptr(d_eye), hpx.ptr(d_inv),
                                           hpx.ptr(info), hpx.stream_ptr(torch)), "hpx_zpotrs_batched")
    return d_inv if int(info.item()) == 0 else None


def _is_hpd_candidate(Fm):
    Fm = np.asarray(Fm)
    return Fm.ndim == 2 and np.allclose(Fm, Fm.conj().T, rtol=1e-13, atol=0) and np.all(np.diag(Fm).real > 0)


def M_Finv(Fm):
    """``inv(F)`` (oqe.py:73-74).  A Hermitian Fisher matrix (any Hermitian weighting ``R`` gives one)
    is inverted on the GPU by the batched Cholesky solver; other matrices fall back to LAPACK."""
    if _is_hpd_candidate(Fm):
        torch, dev = _dev()
        d_inv = _device_inverse(torch, dev, _c128(torch, dev, np.asarray(Fm)[None]))
        if d_inv is not None:
            out = d_inv[0].cpu().numpy()
            return out.real.copy() if np.isrealobj(Fm) else out
    return np.linalg.inv(Fm)


def M_Fhalf(Fm, tol=1e-9, maxit=80):
    """``inv(sqrtm(F))`` (oqe.py:69-70).  For a Hermitian positive-definite ``F`` the principal inverse square root
    comes from ONE library call: the coupled Newton-Schulz iteration of ``hpx_sqrtm_hpd_batched`` (batched FP64-MFMA
    products, the stop test on the device side of the call) returns it directly.  ``F`` is padded to a multiple of 16
    with its mean diagonal value.  Other matrices -- and any result that is not a solution to 1e-6 -- fall back to
    scipy's ``sqrtm``."""
    if _is_hpd_candidate(Fm):
        torch, dev = _dev()
        Fh = np.asarray(Fm, dtype=complex)
        s = Fh.shape[0]
        npad = 16 * ((s + 15) // 16)
        A = np.zeros((1, npad, npad), dtype=complex)
        A[0, :s, :s] = Fh
        if npad > s:
            A[0, np.arange(s, npad), np.arange(s, npad)] = float(np.trace(Fh).real) / s
        d_A = _c128(torch, dev, A)
        d_Z = torch.empty_like(d_A)
        rc = hpx.lib().hpx_sqrtm_hpd_batched(1, npad, hpx.ptr(d_A), None, hpx.ptr(d_Z), float(tol), int(maxit), None,
                                             hpx.stream_ptr(torch))
        if rc == hpx.HPX_OK:
            out = d_Z[0, :s, :s].cpu().numpy()
            # accept only a solution: Z F Z = I to rounding x condition (one host product)
            resid = np.abs(out @ Fh @ out - np.eye(s)).max()
            if np.isfinite(resid) and resid < 1e-6:
                return out.real.copy() if np.isrealobj(Fm) else out
    return np.linalg.inv(scipy.linalg.sqrtm(Fm))


def M_opt(Fm):
    """oqe.py:77-84: ``diag(1/F_aa)`` with row ``a`` divided by ``sum_b (M F)_ab`` (HIP kernel)."""
    torch, dev = _dev()
    Fm = np.asarray(Fm)
    d_F = _c128(torch, dev, Fm[None])
    d_M = torch.empty_like(d_F)
    hpx.check(hpx.lib().hpx_oqe_mopt(1, Fm.shape[0], hpx.ptr(d_F), hpx.ptr(d_M), hpx.stream_ptr(torch)),
              "hpx_oqe_mopt")
    out = d_M[0].cpu().numpy()
    return out.real.copy() if np.isrealobj(Fm) else out


def _sandwich_diag(R, Cm, conj_right):
    """diag(M (R C R') M^H), R' = conj(R) or R, on the device (three dense products + a row dot)."""
    torch, dev = _dev()
    s = np.shape(R)[0]
    d_R, d_C = _c128(torch, dev, np.asarray(R)[None]), _c128(torch, dev, np.asarray(Cm)[None])
    d_o = torch.empty((1, s), dtype=torch.complex128, device=dev)
    hpx.check(hpx.lib().hpx_oqe_sandwich_diag(1, s, hpx.ptr(d_R), hpx.ptr(d_C), int(conj_right), hpx.ptr(d_o),
                                              None, 0, hpx.stream_ptr(torch)), "hpx_oqe_sandwich_diag")
    return d_o[0].cpu().numpy()


def bias_all(s, R, C_noise_total):
    """The noise bias for every delay at once: ``1/2 diag(M (R C conj R) M^H)``."""
    return 0.5 * _sandwich_diag(R, C_noise_total, True)


_bias_cache = {"key": None, "val": None}


def bias(tau, s, R, C_noise_total):  # oqe.py:23-24
    """One delay of :func:`bias_all`.  The reference is called once per delay (oqe.py:88-101 loops over tau):
    the O(s^3) sandwich of the last (R, C) pair is kept, keyed by the arrays' contents, so that such a loop
    costs one device pass instead of s."""
    import hashlib
    key = (s, hashlib.sha1(np.ascontiguousarray(R).tobytes()).digest(),
           hashlib.sha1(np.ascontiguousarray(C_noise_total).tobytes()).digest())
    if _bias_cache["key"] != key:
        _bias_cache["val"] = bias_all(s, R, C_noise_total)
        _bias_cache["key"] = key
    return _bias_cache["val"][tau]


def _q_auto(V, s, R):
    """1/2 x^H (R^* Q_tau R) x for every row x of V and every tau, on the device (closed form
    1/2 conj(FFT(R^T x)) FFT(R x): Q_tau is the rank-one outer(conj m_tau, m_tau))."""
    torch = hpx.require_gpu()
    V = np.atleast_2d(np.asarray(V, dtype=complex))
    dev = torch.device("cuda", torch.cuda.current_device())
    d_R = hpx.to_dev(torch, np.asarray(R, dtype=complex)[None], torch.complex128, dev)
    d_V = hpx.to_dev(torch, V[None], torch.complex128, dev)
    d_q = torch.zeros((1, len(V), s), dtype=torch.complex128, device=dev)
    hpx.check(hpx.lib().hpx_oqe_qauto(1, len(V), s, hpx.ptr(d_R), hpx.ptr(d_V), hpx.ptr(d_q), None, 0,
                                      hpx.stream_ptr(torch)), "hpx_oqe_qauto")
    return d_q[0].cpu().numpy()


def qhat(x, tau, s, R, bias):       # oqe.py:27-30
    return _q_auto(np.asarray(x).reshape(1, s), s, R)[0, tau] - bias


def q_h(V, s, R, taper=None):
    """Cross-correlation estimator over consecutive visibility pairs (oqe.py:104-114):
    ``V`` (2P,s) -> (P,s) complex."""
    torch = hpx.require_gpu()
    V = np.asarray(V, dtype=complex)
    npair = len(V) // 2
    dev = torch.device("cuda", torch.cuda.current_device())
    d_R = hpx.to_dev(torch, np.asarray(R, dtype=complex)[None], torch.complex128, dev)
    d_V = hpx.to_dev(torch, V[None, :2 * npair], torch.complex128, dev)
    d_q = torch.zeros((1, npair, s), dtype=torch.complex128, device=dev)
    hpx.check(hpx.lib().hpx_oqe_qh(1, npair, s, hpx.ptr(d_R), hpx.ptr(d_V), hpx.ptr(d_q), None, 0,
                                   hpx.stream_ptr(torch)), "hpx_oqe_qh")
    return d_q[0].cpu().numpy()


def qhat_h(x1, x2, tau, s, R):      # oqe.py:33-40
    return q_h(np.stack([x1, x2]), s, R)[0, tau]


def q_hp(V, s, R, ncpu):            # oqe.py:147-158 (the process pool is not needed)
    return list(q_h(V, s, R))


def q(V, s, R, bias):               # oqe.py:88-101 (real array, as the reference allocates it)
    return (_q_auto(V, s, R) - np.asarray(bias)[None, :]).real.copy()


def p(q, M):                        # oqe.py:117-118
    return M @ q


def matc(M):                        # oqe.py:121-127
    evs = np.linalg.eigvals(M).real
    Minv = np.linalg.inv(M)
    print(np.all(evs > 0), ' - positive definite')
    print(np.format_float_scientific(max(evs) / min(evs)), ' - eigval ratio')
    print('%f' % (np.linalg.norm(M) * np.linalg.norm(Minv)), ' - condition (norm C x norm Cinv)')
    print('')


def getqs(Vis, R):                  # oqe.py:130-144
    s = len(Vis[0])
    matc(R)
    Fm = F(s, R)
    return q_h(Vis, s, R), Fm, M_opt(Fm), M_Finv(Fm)


def Sig_QEN(R, C_noise, norm):      # oqe.py:161-173
    """``E_i = norm R Q_i R`` is rank one (``u_i v_i^T``), so ``1/2 tr(E C E C) = 1/2 norm^2 n_i^2`` with
    ``n = diag(M (R C R) M^H)`` (device)."""
    assert np.ndim(norm) == 0, "Sig_QEN: `norm` must be a scalar (the rank-one identity does not hold for an array)"
    n = _sandwich_diag(R, C_noise, False)
    return 0.5 * norm ** 2 * n * n


def Sig_QESN(R, C_noise, C_S, norm):  # oqe.py:177-186
    assert np.ndim(norm) == 0, "Sig_QESN: `norm` must be a scalar (the rank-one identity does not hold for an array)"
    n = _sandwich_diag(R, C_noise, False)
    sg = _sandwich_diag(R, C_S, False)
    return 0.5 * norm ** 2 * (n * n + 2.0 * sg * n)
