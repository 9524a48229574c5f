"""Weighted DPSS fit to masked complex spectra (mirror of ``hydra_pspec.dpss``,
reference hydra_pspec/dpss.py:7-94).

The reference minimises the quadratic ``0.5 x^H C^-1 x``,
``x = taper*w*(d - sum_k c_k mode_k)``, with L-BFGS-B from a zero start.  The HIP
path solves the same problem in closed form (normal equations in the weighted
basis), grouped by the spectra that share their weights: per group the N x N contraction
``C^-1 (tw * modes^T)``, the weighted normal matrix and its inverse (`hpx_dpss_project_grouped`, dense
product on FP64 MFMA); per spectrum the tall-skinny projection and the ``nmodes x nmodes``
multiply in one MFMA kernel that reads the visibilities once.  The result differs from the reference only by the
optimiser's stopping slack (it is the exact minimiser; tests check that its cost is
never larger).  ``dpss_fit_modes_batched`` fits many spectra that share ``cov`` --
the (baseline x time) batch of the north star.
"""
import numpy as np
from scipy.signal.windows import dpss

from . import hpx


def inverse_covariance_device(torch, cov, dev):
    """``inv(cov)`` (reference dpss.py:75) as a device tensor.  Hermitian positive-definite matrices
    (every covariance is one) are inverted on the GPU through the batched Cholesky solver with the
    identity as right-hand side; anything else falls back to LAPACK on the host, as the reference."""
    cov = np.asarray(cov, dtype=complex)
    N = cov.shape[0]
    herm = np.array_equal(cov, cov.conj().T)
    if herm:
        d_cov = hpx.to_dev(torch, cov[None], torch.complex128, dev)
        d_eye = torch.eye(N, dtype=torch.complex128, device=dev)[None].contiguous()
        d_inv = torch.empty((1, N, N), dtype=torch.complex128, device=dev)
        info = torch.zeros(1, dtype=torch.int32, device=dev)
        hpx.check(hpx.lib().hpx_zpotrs_batched(1, N, N, hpx.ptr(d_cov), hpx.ptr(d_eye), hpx.ptr(d_inv),
                                               hpx.ptr(info), hpx.stream_ptr(torch)), "hpx_zpotrs_batched")
        if int(info.item()) == 0:
            return d_inv[0]
    return hpx.to_dev(torch, np.linalg.inv(cov), torch.complex128, dev)


class DpssProjector:
    """Workspace and device-resident operands of the grouped DPSS fit: build once, call per batch
    (nothing is allocated and nothing synchronises inside :meth:`fit`)."""

    def __init__(self, ngroups, per, freqs, cov, nmodes=10, alpha=1., taper=None, device=None):
        torch = hpx.require_gpu()
        self.torch = torch
        self.dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.ng, self.per, self.N, self.nm = int(ngroups), int(per), int(freqs.size), int(nmodes)
        assert cov.shape == (self.N, self.N), \
            "Data, flags, covariance, and freqs arrays must have same number of channels"
        if taper is not None:
            assert taper.size == freqs.size, "'taper' must be evaluated at locations given in 'freqs'"
        self.taper = 1. if taper is None else np.asarray(taper, dtype=float)
        self.modes = dpss(freqs.size, NW=alpha, Kmax=nmodes, sym=False)
        self.d_modes = hpx.to_dev(torch, self.modes, torch.float64, self.dev)
        self.d_icov = inverse_covariance_device(torch, cov, self.dev)
        nbytes = int(hpx.lib().hpx_dpss_workspace_bytes(self.ng, self.per, self.N, self.nm))
        self.work = torch.empty(nbytes // 8 + 1, dtype=torch.float64, device=self.dev)
        self.work_bytes = nbytes
        self._have_projector = False

    def fit(self, d, w=None, out=None):
        """``d`` (ngroups, per, N) complex (host or device), ``w`` (ngroups, N) weights shared by the
        spectra of a group; returns the device tensor of amplitudes (ngroups, per, 2*nmodes).
        ``w=None``: the weights of the previous call (its projector is reused, only the projection runs)."""
        torch = self.torch
        d_d = hpx.to_dev(torch, d, torch.complex128, self.dev)
        assert tuple(d_d.shape) == (self.ng, self.per, self.N)
        if w is None:
            assert self._have_projector, "fit(d, w=None) needs an earlier call with weights"
            d_tw = None
        elif isinstance(w, np.ndarray):
            w = np.ascontiguousarray(np.broadcast_to(w, (self.ng, self.N)) * self.taper, dtype=float)
            d_tw = hpx.to_dev(torch, w, torch.float64, self.dev)
        else:
            d_tw = (w.to(self.dev, torch.float64) * hpx.to_dev(torch, np.broadcast_to(self.taper, (self.N,)).copy(),
                                                            torch.float64, self.dev)).contiguous()
        if out is None:
            out = torch.empty((self.ng, self.per, 2 * self.nm), dtype=torch.float64, device=self.dev)
        hpx.check(hpx.lib().hpx_dpss_project_grouped(
            self.ng, self.per, self.N, self.nm, hpx.ptr(d_d), hpx.ptr(d_tw), hpx.ptr(self.d_modes),
            hpx.ptr(self.d_icov), hpx.ptr(out), hpx.ptr(self.work), self.work_bytes, int(w is None),
            hpx.stream_ptr(torch)), "hpx_dpss_project_grouped")
        self._have_projector = True
        return out

    def singular_groups(self):
        """Indices of the groups whose weighted normal matrix was not positive definite in the last fit with
        weights (fully flagged, or fewer unflagged channels than modes): their amplitudes are zero.
        Synchronises the stream."""
        info = np.zeros(self.ng, dtype=np.int32)
        hpx.check(hpx.lib().hpx_dpss_group_info(hpx.ptr(self.work), self.work_bytes, self.ng, self.per, self.N,
                                                self.nm, info.ctypes.data, hpx.stream_ptr(self.torch)),
                  "hpx_dpss_group_info")
        return np.flatnonzero(info)


def dpss_fit_modes_batched(d, w, freqs, cov, nmodes=10, alpha=1., taper=None):
    """Many spectra that share ``cov``.  ``d`` (nb,N) complex with ``w`` (nb,N) (every spectrum its own
    weights) or (N,) (one set for all), or ``d`` (ngroups,per,N) with ``w`` (ngroups,N) -- the
    (baseline x time) cube of the north star, weights shared by the times of a baseline.
    Returns (modes (nmodes,N), amps (.., 2*nmodes))."""
    d = np.asarray(d, dtype=complex) if isinstance(d, np.ndarray) or not hasattr(d, "detach") else d
    if d.ndim == 1:
        d = d[None]
    w = np.asarray(w, dtype=float)
    N = d.shape[-1]
    assert N == cov.shape[0] == cov.shape[1] == freqs.size and w.shape[-1] == N, \
        "Data, flags, covariance, and freqs arrays must have same number of channels"
    if d.ndim == 3:
        ng, per = d.shape[:2]
        shape = (ng, per)
    elif w.ndim == 1:
        ng, per, shape = 1, d.shape[0], (d.shape[0],)
    else:
        ng, per, shape = d.shape[0], 1, (d.shape[0],)
    pr = DpssProjector(ng, per, freqs, cov, nmodes=nmodes, alpha=alpha, taper=taper)
    out = pr.fit(np.ascontiguousarray(d).reshape(ng, per, N), np.broadcast_to(w, (ng, N)))
    bad = pr.singular_groups()
    if bad.size:
        import warnings
        warnings.warn(f"dpss fit: the weighted normal matrix of {bad.size} group(s) (first: {int(bad[0])}) is not "
                      "positive definite (fully flagged, or fewer unflagged channels than modes); their amplitudes "
                      "are zero", RuntimeWarning, stacklevel=2)
    return pr.modes, out.cpu().numpy().reshape(shape + (2 * nmodes,))


def dpss_fit_modes(d, w, freqs, cov, nmodes=10, alpha=1., minimize_method='L-BFGS-B', taper=None):
    """Same signature and return value as the reference (dpss.py:7): ``(dpss_modes
    (nmodes,N), amps (2*nmodes,) real/imag interleaved)``.  ``minimize_method`` is
    accepted and ignored (closed-form solution)."""
    assert d.size == cov.shape[0] == cov.shape[1] == freqs.size == w.size, \
        "Data, flags, covariance, and freqs arrays must have same number of channels"
    modes, amps = dpss_fit_modes_batched(np.asarray(d)[None], np.asarray(w)[None], freqs, cov,
                                         nmodes=nmodes, alpha=alpha, taper=taper)
    return modes, amps[0]
