"""Restatement of the reference DPSS fit (TEST INFRASTRUCTURE, see oracle/__init__.py).

``dpss_fit_modes`` follows dpss.py:7-94: DPSS basis from
``scipy.signal.windows.dpss(N, NW=alpha, Kmax=nmodes, sym=False)`` (:69-72),
``inv(cov)`` (:75), then L-BFGS-B minimisation of the quadratic
``0.5 x^H C^-1 x`` with ``x = taper*w*(d - sum_k c_k mode_k)`` from a zero start
(:78-92).  ``dpss_fit_closed_form`` is the same minimum written as normal
equations; it is what the HIP kernel computes and is checked against the
optimiser in tests (they differ by the optimiser's stopping tolerance only).
"""
import numpy as np
from scipy.optimize import minimize
from scipy.signal.windows import dpss


def dpss_fit_modes(d, w, freqs, cov, nmodes=10, alpha=1., minimize_method="L-BFGS-B",
                   taper=None):
    assert d.size == cov.shape[0] == cov.shape[1] == freqs.size == w.size, \
        "Data, flags, covariance, and freqs arrays must have same number of channels"
    if taper is None:
        taper = 1.
    else:
        assert taper.size == freqs.size, "'taper' must be evaluated at locations given in 'freqs'"
    modes = dpss(freqs.size, NW=alpha, Kmax=nmodes, sym=False)
    icov = np.linalg.inv(cov)

    def objective(p):
        m = np.sum(p[0::2, None] * modes + 1.j * p[1::2, None] * modes, axis=0)
        x = taper * w * (d - m)
        return (0.5 * np.dot(x.conj(), np.dot(icov, x))).real

    res = minimize(objective, np.zeros(2 * nmodes), method=minimize_method, bounds=None)
    return modes, res.x


def dpss_fit_closed_form(d, w, freqs, cov, nmodes=10, alpha=1., taper=None):
    """argmin of the same quadratic: real normal equations in the 2*nmodes
    interleaved (re, im) unknowns.  For Hermitian C^-1 the cost is
    ``(y-Bc)^H Ci (y-Bc)`` with B = diag(taper*w) modes^T (real), c complex, so
    ``(B^T Ci B) c = B^T Ci y`` (complex Hermitian system); for a general
    (non-Hermitian) ``inv(cov)`` only its Hermitian part enters the real cost."""
    if taper is None:
        taper = 1.
    modes = dpss(freqs.size, NW=alpha, Kmax=nmodes, sym=False)
    icov = np.linalg.inv(cov)
    icov_h = 0.5 * (icov + icov.conj().T)
    tw = np.asarray(taper * w, dtype=float) * np.ones(freqs.size)
    B = (modes * tw[None, :]).T                       # (N, nmodes) real
    y = tw * d
    lhs = B.T @ icov_h @ B
    rhs = B.T @ (icov_h @ y)
    c = np.linalg.solve(lhs, rhs)
    amps = np.empty(2 * nmodes)
    amps[0::2], amps[1::2] = c.real, c.imag
    return modes, amps
