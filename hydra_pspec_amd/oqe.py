"""Optimal-quadratic-estimator helpers (mirror of ``hydra_pspec.oqe``, reference
hydra_pspec/oqe.py).  Same names and return values; the shipped reference file
cannot run as is (it uses ``os``, ``sp`` and ``time`` without importing them and
caches ``Q`` matrices under ``Qs/``), so "reference behaviour" means the behaviour
with those names provided (tests/golden/make_golden.py).

The two expensive pieces run on the GPU: the Fisher matrices ``F`` / ``Ft``
(O(s^5) / O(s^4) trace loops in the reference, here two s x s x s contractions:
``X = M R M^H``, ``Wm = conj(M) R M^T``, ``F_ab = conj(Wm_ba) X_ab / 2``,
``Ft_ab = |X_ab|^2 / 2`` -- valid for any ``R``) and the cross-estimator ``q_h``
(``q_t = conj(FFT(R x1))_t FFT(R x2)_t / 2``).  The remaining helpers are s x s
glue on the host.
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


def _fisher(R, variant):
    torch = hpx.require_gpu()
    R = np.asarray(R, dtype=complex)
    batched = R.ndim == 3
    Rb = R if batched else R[None]
    nb, s, _ = Rb.shape
    dev = torch.device("cuda", torch.cuda.current_device())
    d_R = hpx.to_dev(torch, Rb, torch.complex128, dev)
    d_F = torch.zeros_like(d_R)
    hpx.check(hpx.lib().hpx_oqe_fisher(nb, s, hpx.ptr(d_R), hpx.ptr(d_F), variant, hpx.stream_ptr(torch)),
              "hpx_oqe_fisher")
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


def M_Fhalf(Fm):                    # oqe.py:69-70
    return np.linalg.inv(scipy.linalg.sqrtm(Fm))


def M_Finv(Fm):                     # oqe.py:73-74
    return np.linalg.inv(Fm)


def M_opt(Fm):                      # oqe.py:77-84
    M = np.diag(np.divide(1, np.diag(Fm)))
    W = M @ Fm
    for row in range(M.shape[0]):
        M[row] = np.divide(M[row], np.sum(W[row]))
    return M


def bias(tau, s, R, C_noise_total):  # oqe.py:23-24
    return 0.5 * np.trace(C_noise_total @ R.conj() @ Q(tau, s) @ R)


def _q_auto(V, s, R):
    """1/2 x^H (R^* Q_tau R) x for every row x of V and every tau, on the device (closed form
    1/2 conj(FFT(R^T x)) FFT(R x): Q_tau is the rank-one outer(conj m_tau, m_tau))."""
    torch = hpx.require_gpu()
    V = np.atleast_2d(np.asarray(V, dtype=complex))
    dev = torch.device("cuda", torch.cuda.current_device())
    d_R = hpx.to_dev(torch, np.asarray(R, dtype=complex)[None], torch.complex128, dev)
    d_V = hpx.to_dev(torch, V[None], torch.complex128, dev)
    d_q = torch.zeros((1, len(V), s), dtype=torch.complex128, device=dev)
    hpx.check(hpx.lib().hpx_oqe_qauto(1, len(V), s, hpx.ptr(d_R), hpx.ptr(d_V), hpx.ptr(d_q),
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
    hpx.check(hpx.lib().hpx_oqe_qh(1, npair, s, hpx.ptr(d_R), hpx.ptr(d_V), hpx.ptr(d_q),
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
    s = len(R)
    out = np.zeros(s, dtype=complex)
    for i in range(s):
        E = R @ Q(i, s) @ R * norm
        out[i] = 0.5 * np.trace(E @ C_noise @ E @ C_noise)
    return out


def Sig_QESN(R, C_noise, C_S, norm):  # oqe.py:177-186
    s = len(R)
    out = np.zeros(s, dtype=complex)
    for i in range(s):
        E = R @ Q(i, s) @ R * norm
        out[i] = 0.5 * np.trace((E @ C_noise @ E @ C_noise) + (E @ C_S @ E @ C_noise)
                                + (E @ C_noise @ E @ C_S))
    return out
