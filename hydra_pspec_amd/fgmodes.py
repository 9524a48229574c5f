"""Foreground modes from the data, on the GPU.

The reference derives the modes offline (scripts/calc-vis-cov-matrices.py:235-249):
``cov = np.cov(bl_data.T)`` over the times of a foreground-only simulation, its
eigenvectors saved per baseline, and the driver keeps the first ``Nfgmodes`` columns
(run-hydra-pspec.py:453).  :func:`cov_eig_modes` returns those leading eigenvectors for a
whole batch of baselines at once (``hpx_fgmodes_eig``: batched Hermitian Jacobi
eigensolver, through the Ntimes x Ntimes Gram matrix when Ntimes <= Nfreqs).
"""
import numpy as np

from . import hpx


def cov_eig_modes(vis, nmodes, return_evals=False, as_numpy=True):
    """``vis`` (Nbl,Ntimes,Nfreqs) or (Ntimes,Nfreqs) complex -> modes (Nbl,Nfreqs,nmodes)
    [(Nfreqs,nmodes) for a single baseline]: unit-norm eigenvectors of ``np.cov(vis_b.T)`` for
    the ``nmodes`` largest eigenvalues in descending order, each with its largest component
    real and positive (an eigenvector's phase is arbitrary; the sampler only uses the span)."""
    torch = hpx.require_gpu()
    single = (len(vis.shape) == 2)
    if single:
        vis = vis[None]
    nbl, T, N = tuple(vis.shape)
    if min(T, N) > 1024:
        raise NotImplementedError("cov_eig_modes needs min(Ntimes, Nfreqs) <= 1024")
    if not 0 < nmodes <= min(T - 1, N):
        raise ValueError("nmodes must be between 1 and min(Ntimes - 1, Nfreqs) (the rank of the covariance)")
    dev = torch.device("cuda", torch.cuda.current_device())
    d_vis = hpx.to_dev(torch, vis, torch.complex128, dev)
    modes = torch.empty((nbl, N, nmodes), dtype=torch.complex128, device=dev)
    evals = torch.empty((nbl, nmodes), dtype=torch.float64, device=dev)
    hpx.check(hpx.lib().hpx_fgmodes_eig(nbl, T, N, nmodes, hpx.ptr(d_vis), hpx.ptr(modes), hpx.ptr(evals),
                                        hpx.stream_ptr(torch)), "hpx_fgmodes_eig")
    if single:
        modes, evals = modes[0], evals[0]
    if as_numpy:
        modes, evals = modes.cpu().numpy(), evals.cpu().numpy()
    return (modes, evals) if return_evals else modes
