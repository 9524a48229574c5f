"""Restatement of the reference OQE helpers (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows oqe.py line by line, with the two defects of the shipped file removed
so that it runs: ``Q`` (oqe.py:13-20) is computed directly instead of being
cached under ``Qs/`` (the reference needs ``os`` which it never imports), and
``M_Fhalf`` (oqe.py:69-70) uses ``scipy.linalg`` explicitly (the reference's
``sp`` is undefined).  The loops are kept (O(s^5) Fisher) -- this is the
checker, not the fast path.
"""
import numpy as np
import scipy.linalg


def m(tau, s):                      # oqe.py:7-10
    y = np.zeros(s)
    y[tau] = 1
    return np.fft.fft(y)


def Q(tau, s):                      # oqe.py:13-20 (no disk cache)
    v = m(tau, s)
    return np.outer(v.conj(), v)


def bias(tau, s, R, C_noise_total):  # oqe.py:23-24
    return 0.5 * np.trace(C_noise_total @ R.conj() @ Q(tau, s) @ R)


def qhat(x, tau, s, R, bias):       # oqe.py:27-30
    E = R.conj() @ Q(tau, s) @ R
    return 0.5 * (x.conj().T @ E @ x) - bias


def qhat_h(x1, x2, tau, s, R):      # oqe.py:33-40
    Rx1, Rx2 = R @ x1, R @ x2
    return 0.5 * Rx1.conj().T @ Q(tau, s) @ Rx2


def F(s, R):                        # oqe.py:43-50
    out = np.zeros((s, s), dtype=complex)
    for a in range(s):
        for b in range(s):
            out[a, b] = 0.5 * np.trace(R.conj() @ Q(a, s) @ R @ Q(b, s))
    return out


def Ft(s, R):                       # oqe.py:53-66
    left = [np.dot(np.conj(R).T, Q(i, s)) for i in range(s)]
    right = [np.dot(R, Q(i, s)) for i in range(s)]
    out = np.zeros((s, s), dtype=complex)
    for a in range(s):
        for b in range(s):
            out[a, b] = 0.5 * np.einsum("ab,ba", left[a], right[b])
    return out


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


def q(V, s, R, bias):               # oqe.py:88-101
    out = np.zeros((len(V), s))
    for i in range(len(V)):
        out[i] = np.array([qhat(V[i], tau, s, R, bias[tau]) for tau in range(s)])
    return out


def q_h(V, s, R, taper=None):       # oqe.py:104-114
    npair = len(V) // 2
    out = np.zeros((npair, s), dtype=complex)
    for i in range(npair):
        out[i] = np.array([qhat_h(V[2 * i], V[2 * i + 1], t, s, R) for t in range(s)])
    return out


def p(q, M):                        # oqe.py:117-118
    return M @ q


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


def getqs(Vis, R):                  # oqe.py:130-144 (prints dropped)
    s = len(Vis[0])
    Fm = F(s, R)
    return q_h(Vis, s, R), Fm, M_opt(Fm), M_Finv(Fm)


# ---- closed forms (Q_tau is rank one): what the device computes, usable at sizes where the loops above
# cannot run; tests/test_oracle_golden.py pins them to the loop forms at small s.
def _M(s):
    return np.fft.fft(np.eye(s))         # row tau = m(tau, s)


def F_closed(s, R):
    Mm = _M(s)
    X = Mm @ R @ Mm.conj().T
    Wm = Mm.conj() @ R @ Mm.T
    return 0.5 * Wm.T.conj() * X


def Ft_closed(s, R):
    Mm = _M(s)
    return 0.5 * np.abs(Mm @ R @ Mm.conj().T) ** 2


def q_h_closed(V, s, R):
    Y = np.fft.fft(np.asarray(V) @ R.T, axis=-1)          # rows: FFT(R x)
    return 0.5 * Y[0::2].conj() * Y[1::2]
