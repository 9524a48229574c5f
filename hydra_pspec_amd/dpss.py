"""Weighted DPSS fit to masked complex spectra (mirror of ``hydra_pspec.dpss``,
reference hydra_pspec/dpss.py:7-94).

The reference minimises the quadratic ``0.5 x^H C^-1 x``,
``x = taper*w*(d - sum_k c_k mode_k)``, with L-BFGS-B from a zero start.  The HIP
path solves the same problem in closed form (normal equations in the weighted
basis): the N x N contraction ``C^-1 (tw * [modes^T | d])`` runs on FP64 MFMA
(`hpx_dpss_project`), followed by the tall-skinny projection and an
``nmodes x nmodes`` solve.  The result differs from the reference only by the
optimiser's stopping slack (it is the exact minimiser; tests check that its cost is
never larger).  ``dpss_fit_modes_batched`` fits many spectra that share ``cov`` --
the (baseline x time) batch of the north star.
"""
import numpy as np
from scipy.signal.windows import dpss

from . import hpx


def dpss_fit_modes_batched(d, w, freqs, cov, nmodes=10, alpha=1., taper=None):
    """``d`` (nb,N) complex, ``w`` (nb,N) or (N,) weights; returns (modes (nmodes,N), amps (nb,2*nmodes))."""
    torch = hpx.require_gpu()
    d = np.atleast_2d(np.asarray(d, dtype=complex))
    nb, N = d.shape
    w = np.broadcast_to(np.asarray(w, dtype=float), (nb, N))
    assert N == cov.shape[0] == cov.shape[1] == freqs.size, \
        "Data, flags, covariance, and freqs arrays must have same number of channels"
    if taper is None:
        taper = 1.
    else:
        assert taper.size == freqs.size, "'taper' must be evaluated at locations given in 'freqs'"
    modes = dpss(freqs.size, NW=alpha, Kmax=nmodes, sym=False)
    icov = np.linalg.inv(cov)
    tw = np.ascontiguousarray(w * taper, dtype=float)
    dev = torch.device("cuda", torch.cuda.current_device())
    f64, c128 = torch.float64, torch.complex128
    d_d = hpx.to_dev(torch, d, c128, dev)
    d_tw = hpx.to_dev(torch, tw, f64, dev)
    d_m = hpx.to_dev(torch, modes, f64, dev)
    d_ic = hpx.to_dev(torch, icov.astype(complex), c128, dev)
    d_out = torch.zeros((nb, 2 * nmodes), dtype=f64, device=dev)
    hpx.check(hpx.lib().hpx_dpss_project(nb, N, int(nmodes), hpx.ptr(d_d), hpx.ptr(d_tw), hpx.ptr(d_m),
                                         hpx.ptr(d_ic), hpx.ptr(d_out), hpx.stream_ptr(torch)),
              "hpx_dpss_project")
    return modes, d_out.cpu().numpy()


def dpss_fit_modes(d, w, freqs, cov, nmodes=10, alpha=1., minimize_method='L-BFGS-B', taper=None):
    """Same signature and return value as the reference (dpss.py:7): ``(dpss_modes
    (nmodes,N), amps (2*nmodes,) real/imag interleaved)``.  ``minimize_method`` is
    accepted and ignored (closed-form solution)."""
    assert d.size == cov.shape[0] == cov.shape[1] == freqs.size == w.size, \
        "Data, flags, covariance, and freqs arrays must have same number of channels"
    modes, amps = dpss_fit_modes_batched(np.asarray(d)[None], np.asarray(w)[None], freqs, cov,
                                         nmodes=nmodes, alpha=alpha, taper=taper)
    return modes, amps[0]
