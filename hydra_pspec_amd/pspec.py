"""Gibbs sampler for the EoR bandpowers with a foreground-mode model: host side.

Mirror of ``hydra_pspec.pspec`` (reference hydra_pspec/pspec.py) for the
MI355X path: same function names, argument meaning, return shapes and error
types.  The arithmetic is done by the HIP kernels in ``csrc/`` through the
C-ABI of ``include/hpx.h``; this module only prepares inputs (including the
reference's random-number streams, reproduced bit for bit with numpy's legacy
MT19937) and lays out outputs.

Formulation: with ``S = U diag(ps/N) U^H`` (``U = F^H/sqrt N``)
and diagonal ``Ninv``, the reference's non-Hermitian system ``A x = b``
(pspec.py:365-369) is solved as the Hermitian positive-definite system
``K' [y'; f] = r'`` in the delay basis -- on the device in the symmetrically scaled form
``M [z; f] = A^-1 r'``, ``z = a . y'`` -- by a batched Cholesky factorisation (or, for unflagged
flat-noise data, through the diagonal + rank-Nmodes structure of ``M``); ``s = U z``.  The accelerated entry point is
:func:`gibbs_sample_with_fg_batched`; :func:`gibbs_sample_with_fg` is its
``Nbl = 1`` drop-in special case.
"""
import ctypes as C
import os
import time

import numpy as np
import scipy.special

from . import hpx, utils

GCR_SEED0 = 912983          # reference pspec.py:153 (multiprocess_seed's default)
NGRID = 1000                # reference pspec.py:11 (ngrid default)
FOURIER_FORM_TOL = 1e-9     # relative off-diagonal power allowed in F S F^H


# --------------------------------------------------------------------------- RNG
def omega_table(T, N, idx=None, seed0=GCR_SEED0):
    """The reference's GCR noise draws: for time ``t`` the legacy stream seeded
    with ``912983 + t`` yields ``omi, omj, omk, oml = randn(N,1) x 4`` in that
    order (pspec.py:196-216).  Identical for every iteration and baseline.
    ``idx`` selects explicit time indices (default ``range(T)``).
    Returns (T,4,N) float64."""
    idx = range(T) if idx is None else idx
    out = np.empty((len(idx), 4, N))
    for i, t in enumerate(idx):
        out[i] = np.random.RandomState(seed0 + int(t)).randn(4, N)
    return out


def draw_tables(T, N, Niter, seed, reseed=True):
    """Uniforms the chain's parent stream hands to the bandpower draw: exactly
    one per channel per iteration, in channel order, for both branches of
    ``sample_S`` (pspec.py:58, :125; scipy's ``invgamma.rvs`` is ``ppf(U)``).
    Uses -- and advances -- numpy's GLOBAL legacy stream like the reference
    (``np.random.seed(seed)`` at pspec.py:577, skipped for map_estimate).
    Returns (uniforms, 1/gammainccinv(T-1, uniforms)), each (Niter,N)."""
    if reseed:
        np.random.seed(seed)
    u = np.random.random_sample((Niter, N))
    igy = np.empty_like(u)
    # element-wise (same values however it is split); scipy's ufunc loop runs without the GIL: rows over a few threads
    nthr = min(8, os.cpu_count() or 1, max(1, u.size // 65536))
    if nthr > 1:
        from concurrent.futures import ThreadPoolExecutor
        cuts = np.linspace(0, Niter, nthr + 1).astype(int)
        with ThreadPoolExecutor(nthr) as ex:
            list(ex.map(lambda i: scipy.special.gammainccinv(T - 1.0, u[cuts[i]:cuts[i + 1]], out=igy[cuts[i]:cuts[i + 1]]),
                        range(nthr)))
    else:
        scipy.special.gammainccinv(T - 1.0, u, out=igy)
    np.divide(1.0, igy, out=igy)
    return u, igy


# ------------------------------------------------------------------ input checks
def _prior_tables(ps_prior, N):
    """(.., 2, N) prior box -> (prior_map int32 (.., N), xgrid (nrows, NGRID)).
    Raises the reference's ValueErrors (pspec.py:40-47) for bad bounds."""
    pr = np.asarray(ps_prior, dtype=float)
    assert pr.shape[-2:] == (2, N), "ps_prior must have shape (2, Nfreqs)"
    has = np.any(pr > 0, axis=-2)
    hi, lo = pr[..., 0, :], pr[..., 1, :]
    if np.any(has):
        if np.any(lo[has] <= 0):
            raise ValueError("prior_min must be greater than zero")
        if np.any(hi[has] <= 0):
            raise ValueError("prior_max must be greater than zero")
        if not np.all(np.isfinite(hi[has])):
            raise ValueError("prior_max must be finite")
        if np.any(hi[has] <= lo[has]):
            raise ValueError("prior_max must be greater than prior_min")
    pmap = np.full(has.shape, -1, dtype=np.int32)
    rows, index = [], {}
    for idx in zip(*np.nonzero(has)):
        key = (lo[idx], hi[idx])
        if key not in index:
            index[key] = len(rows)
            rows.append(np.logspace(np.log10(key[0]), np.log10(key[1]), NGRID))
        pmap[idx] = index[key]
    xgrid = np.array(rows) if rows else np.zeros((0, NGRID))
    return pmap, xgrid


def sqrtm_hermitian(A):
    """Principal square root of Hermitian positive semi-definite matrices (.., N, N) through one
    eigendecomposition (what ``scipy.linalg.sqrtm`` returns for them; reference pspec.py:362)."""
    A = np.asarray(A, dtype=complex)
    lam, V = np.linalg.eigh(0.5 * (A + np.conj(np.swapaxes(A, -1, -2))))
    return (V * np.sqrt(np.clip(lam, 0.0, None))[..., None, :]) @ np.conj(np.swapaxes(V, -1, -2))


def sqrtm_masked_device(torch, nd, w, device, tol=1e-7, max_iter=60):
    """``sqrtm(Ninv diag(w))`` for a stack of Hermitian positive-definite ``Ninv`` (K,N,N) -- or one (N,N) matrix shared
    by all -- and channel masks ``w`` (K,N) bool (True = use), on the device: what the reference gets from
    ``scipy.linalg.sqrtm(Ni)`` per baseline (pspec.py:361-362).  With the unflagged channels ``u`` first,
    ``Ni = [[A, 0], [B, 0]]`` with ``A = Ninv[u, u]`` Hermitian positive definite, and the principal root is
    ``[[A^1/2, 0], [B A^-1/2, 0]]``.  One library call (``hpx_sqrtm_masked_batched``: permutation, Newton-Schulz on
    the ``A`` blocks, ``B A^-1/2`` and the scatter back are kernels of this repository, 256 systems at a time); a
    matrix the iteration does not converge for (ill-conditioned beyond ``max_iter`` steps, or only positive
    SEMI-definite) falls back to the host's eigendecomposition for the whole call.
    Returns a (K,N,N) complex128 device tensor."""
    c128 = torch.complex128
    nd = nd if hasattr(nd, "device") and not isinstance(nd, np.ndarray) else hpx.to_dev(torch, np.ascontiguousarray(nd), c128, device)
    w_np = np.ascontiguousarray(np.asarray(w).astype(bool))
    K, N = int(w_np.shape[0]), int(nd.shape[-1])
    shared = nd.dim() == 2
    assert shared or int(nd.shape[0]) == K
    nd = nd.contiguous()
    d_w = hpx.to_dev(torch, w_np.astype(np.uint8), torch.uint8, device)
    out = torch.empty((K, N, N), dtype=c128, device=device)
    rc = hpx.lib().hpx_sqrtm_masked_batched(K, N, hpx.ptr(nd), int(shared), hpx.ptr(d_w), hpx.ptr(out), float(tol),
                                            int(max_iter), None, hpx.stream_ptr(torch))
    if rc == hpx.HPX_EINVAL and "sqrtm" in hpx.last_error():
        # not converged / not positive definite: the exact route on the host (eigh of the unflagged block)
        full = nd.cpu().numpy()
        host = np.zeros((K, N, N), dtype=complex)
        for k in range(K):
            A = full if shared else full[k]
            u, f = np.where(w_np[k])[0], np.where(~w_np[k])[0]
            if len(u) == 0:
                continue
            lam, V = np.linalg.eigh(0.5 * (A[np.ix_(u, u)] + A[np.ix_(u, u)].conj().T))
            if lam.min() <= 0:
                raise FloatingPointError("the inverse noise covariance is not positive definite on the unflagged channels")
            host[k][np.ix_(u, u)] = (V * np.sqrt(lam)) @ V.conj().T
            if len(f):
                host[k][np.ix_(f, u)] = A[np.ix_(f, u)] @ ((V / np.sqrt(lam)) @ V.conj().T)
        return hpx.to_dev(torch, host, c128, device)
    hpx.check(rc, "hpx_sqrtm_masked_batched")
    return out


def _ninv_dense(Ninv, nbl, N):
    """``Ninv`` if it is a non-diagonal (N,N) / (Nbl,N,N) matrix (then it must be Hermitian), else None."""
    if not isinstance(Ninv, np.ndarray) and hasattr(Ninv, "detach"):
        if Ninv.ndim < 2 or tuple(Ninv.shape[-2:]) != (N, N):
            return None
        Ninv = Ninv.detach().cpu().numpy()
    Ninv = np.asarray(Ninv)
    if Ninv.shape not in ((N, N), (nbl, N, N)):
        return None
    d = np.diagonal(Ninv, axis1=-2, axis2=-1)
    if not np.any(Ninv - d[..., None] * np.eye(N) != 0):
        return None
    herm = np.conj(np.swapaxes(Ninv, -1, -2))
    # inv(noise_cov) of an ill-conditioned covariance is Hermitian only to eps * cond (the reference driver passes
    # it unsymmetrised, run-hydra-pspec.py:436): accept that round-off and work with the Hermitian part
    if not np.allclose(Ninv, herm, rtol=0.0, atol=1e-8 * np.abs(Ninv).max()):
        raise NotImplementedError("a non-Hermitian inverse noise covariance is not supported")
    return np.ascontiguousarray(0.5 * (Ninv + herm), dtype=complex)


def _ninv_dense_pertime(Ninv, nbl, T, N):
    """A time-dependent ``Ninv`` (Ntimes,N,N) / (Nbl,Ntimes,N,N) (pspec.py:337-340) with off-diagonal terms ->
    (Nbl,Ntimes,N,N) Hermitian complex; None when every matrix is diagonal (the per-time diagonal mode)."""
    if not isinstance(Ninv, np.ndarray) and hasattr(Ninv, "detach"):
        Ninv = Ninv.detach().cpu().numpy()
    Ninv = np.asarray(Ninv)
    if not (Ninv.shape in ((T, N, N), (nbl, T, N, N)) and (Ninv.ndim == 4 or T != nbl)):
        return None
    d = np.diagonal(Ninv, axis1=-2, axis2=-1)
    if not np.any(Ninv - d[..., None] * np.eye(N) != 0):
        return None
    herm = np.conj(np.swapaxes(Ninv, -1, -2))
    if not np.allclose(Ninv, herm, rtol=0.0, atol=1e-8 * np.abs(Ninv).max()):
        raise NotImplementedError("a non-Hermitian inverse noise covariance is not supported")
    return np.ascontiguousarray(np.broadcast_to(0.5 * (Ninv + herm), (nbl, T, N, N)), dtype=complex)


def _ninv_diag(Ninv, nbl, T, N):
    """Accept (N,), (nbl,N) diagonals or (N,N)/(nbl,N,N) dense matrices that are
    diagonal; return (nbl,N).  Dense non-diagonal inverse covariances make the
    reference's column-masked ``Ni`` non-Hermitian (pspec.py:361 FIXME) and are
    not supported by the Cholesky formulation.  A torch tensor that already is an (nbl,N)
    diagonal stays where it is (device inputs are used in place)."""
    if not isinstance(Ninv, np.ndarray) and hasattr(Ninv, "detach"):
        if tuple(Ninv.shape) == (nbl, N):
            return Ninv
        Ninv = Ninv.detach().cpu().numpy()
    Ninv = np.asarray(Ninv)
    if Ninv.shape in ((T, N, N), (nbl, T, N, N)) and (Ninv.ndim == 4 or T != nbl):
        # time-dependent Ninv (pspec.py:337-340): one diagonal matrix per time -> (nbl, T, N)
        d = np.diagonal(Ninv, axis1=-2, axis2=-1)
        if np.any(Ninv - d[..., None] * np.eye(N) != 0):
            raise NotImplementedError("this entry point takes diagonal time-dependent inverse noise covariances only "
                                      "(make_batch / gibbs_sample_with_fg[_batched] accept Hermitian matrices)")
        return np.ascontiguousarray(np.broadcast_to(d.real, (nbl, T, N)), dtype=float)
    if Ninv.shape == (nbl, T, N) and not (T == N and Ninv.ndim == 3 and nbl == T):
        return np.ascontiguousarray(Ninv.real, dtype=float)
    if Ninv.shape in ((N,), (nbl, N)):
        return np.ascontiguousarray(np.broadcast_to(Ninv.real, (nbl, N)), dtype=float)
    if Ninv.shape in ((N, N), (nbl, N, N)):
        d = np.diagonal(Ninv, axis1=-2, axis2=-1)
        off = Ninv - d[..., None] * np.eye(N)
        if np.any(off != 0):
            raise NotImplementedError("this entry point takes diagonal inverse noise covariances only "
                                      "(make_batch / gibbs_sample_with_fg[_batched] accept Hermitian matrices)")
        return np.ascontiguousarray(np.broadcast_to(d.real, (nbl, N)), dtype=float)
    raise AssertionError("Ninv shape must be (Nfreqs, Nfreqs) or a diagonal (Nfreqs,)")


def pspec_from_covariance(S, fourier_op=None):
    """Inverse of :func:`covariance_from_pspec` for covariances of the form
    ``F^H diag(ps/N^2) F``: returns (ps, relative off-diagonal residual)."""
    S = np.asarray(S)
    N = S.shape[-1]
    F = utils.fourier_operator(N) if fourier_op is None else fourier_op
    D = F @ S @ F.conj().T
    ps = np.diagonal(D, axis1=-2, axis2=-1).real.copy()
    off = D - np.eye(N) * np.diagonal(D, axis1=-2, axis2=-1)[..., None, :]
    resid = np.linalg.norm(off, axis=(-2, -1)) / np.maximum(np.linalg.norm(D, axis=(-2, -1)), 1e-300)
    return ps, resid


def sqrt_cov_delay_basis(S, fourier_op=None):
    """``Sh' = U^H sqrtm(S) U`` (``U = F^H / sqrt N``) for Hermitian positive semi-definite ``S``
    (the principal square root the reference takes with ``scipy.linalg.sqrtm``,
    pspec.py:359), via one Hermitian eigendecomposition on the host.  Used only for
    the first iteration of a chain whose ``S_initial`` is not of the form
    ``F^H diag(.) F``."""
    S = np.asarray(S, dtype=complex)
    N = S.shape[-1]
    F = utils.fourier_operator(N) if fourier_op is None else fourier_op
    Sh = 0.5 * (S + np.conj(np.swapaxes(S, -1, -2)))
    lam, V = np.linalg.eigh(Sh)
    root = (V * np.sqrt(np.clip(lam, 0.0, None))[..., None, :]) @ np.conj(np.swapaxes(V, -1, -2))
    return F @ root @ F.conj().T / N


# ------------------------------------------------------------------ batched core
class GibbsBatch:
    """A batch of independent baselines resident on one GPU.

    Static inputs are uploaded and reduced to the iteration-invariant operators
    once; :meth:`run` then advances all chains by ``niter`` iterations.
    """

    def __init__(self, vis, flags, fgmodes, ninv_diag, ps_prior, Niter, seed=None,
                 map_estimate=False, device=None, tables=None, omega=None, solver="auto", ninv_dense=None,
                 allow_split=True):
        """``allow_split=False``: never the split factorisation (several co-operating workgroups per system, taken for
        batches of fewer baselines than half the CUs): what a caller sets that shares the GPU with other processes, or
        that wants a baseline's chain to be bit for bit the same in a batch of any size (``hpx.OPT_FACTOR_SPLIT``).

        ``ninv_dense``: Hermitian non-diagonal inverse noise covariance(s) (N,N) or (Nbl,N,N) -- or, time
        dependent, (Nbl,Ntimes,N,N) -- instead of
        ``ninv_diag`` (:func:`make_batch` routes them here): dense solver only.  With flagged channels the
        reference's column-masked ``Ni`` is not Hermitian (pspec.py:361): the solution then comes from the
        unflagged-noise factorisation through a rank-f Woodbury correction (``hpx_plan_set_static_dense_flagged``)."""
        torch = hpx.require_gpu()
        self.torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None \
            else torch.device(device)
        vis_shape = tuple(vis.shape)
        assert len(vis_shape) == 3, "vis must have shape (Nbl, Ntimes, Nfreqs)"
        nbl, T, N = vis_shape
        fg_shape = tuple(fgmodes.shape)
        assert fg_shape[-2] == N, "fgmodes must have shape (Nfreqs, Nmodes)"
        M = fg_shape[-1]
        assert tuple(flags.shape) in ((nbl, N), (nbl, T, N)), \
            "`flags` array must have shape (Nbl, Nfreqs) or, time dependent, (Nbl, Ntimes, Nfreqs)"
        dense_t = ninv_dense is not None and np.ndim(ninv_dense) == 4
        if (dense_t or (ninv_dense is None and tuple(np.shape(ninv_diag)) == (nbl, T, N)
                        and not (T == N and nbl == T))) and len(tuple(flags.shape)) == 2:
            # time-dependent noise with time-independent flags: the same flags at every time
            fl2 = flags if isinstance(flags, np.ndarray) else flags.detach().cpu().numpy()
            flags = np.ascontiguousarray(np.broadcast_to(np.asarray(fl2)[:, None, :], (nbl, T, N)))
        self.per_time = len(tuple(flags.shape)) == 3
        self.nbl, self.T, self.N, self.M = nbl, T, N, M
        self.map_estimate = bool(map_estimate)
        self.Niter = 1 if map_estimate else int(Niter)
        with torch.cuda.device(self.device):
            c128, f64 = torch.complex128, torch.float64
            d_vis = hpx.to_dev(torch, vis, c128, self.device)
            fl_np = flags if isinstance(flags, np.ndarray) else flags.detach().cpu().numpy()
            d_flags = hpx.to_dev(torch, np.ascontiguousarray(fl_np).astype(np.uint8), torch.uint8,
                                 self.device)
            self.dense_noise = ninv_dense is not None
            self.any_flags = bool((~fl_np.astype(bool)).any())
            extra_rhs = 0
            if self.dense_noise and self.per_time:
                # one full noise matrix per (baseline, time) (pspec.py:337-340): Ni = Ninv_t diag(w_t) and its
                # scipy sqrtm per unit, as build_matrices would make them for that time (:361-362)
                nd = np.asarray(ninv_dense, dtype=complex)
                if nd.ndim == 3:
                    nd = np.broadcast_to(nd[:, None], (nbl, T, N, N))
                assert nd.shape == (nbl, T, N, N), "time-dependent Ninv must have shape (Nbl, Ntimes, Nfreqs, Nfreqs)"
                nd = np.ascontiguousarray(nd)
                w = fl_np.astype(bool)
                d_nd = hpx.to_dev(torch, nd, c128, self.device)
                d_nh = sqrtm_masked_device(torch, d_nd.reshape(nbl * T, N, N), w.reshape(nbl * T, N),
                                           self.device).reshape(nbl, T, N, N)
                d_ninv = None
            elif self.dense_noise:
                nd = np.asarray(ninv_dense, dtype=complex)
                assert nd.shape in ((N, N), (nbl, N, N)), "Ninv shape must be (Nfreqs, Nfreqs) or (Nbl, Nfreqs, Nfreqs)"
                d_nd = hpx.to_dev(torch, nd, c128, self.device)
                if self.any_flags:
                    # the reference masks the COLUMNS of Ninv (Ni = flags.T * Ninv * flags, pspec.py:361) and takes
                    # scipy's sqrtm of that general matrix (:362): the same principal root, on the device
                    w = fl_np.astype(bool)
                    d_nh = sqrtm_masked_device(torch, d_nd, w, self.device)
                    extra_rhs = int((~w).sum(axis=1).max())        # one more right-hand side per flagged channel
                else:
                    d_nh = sqrtm_masked_device(torch, d_nd.reshape(-1, N, N), np.ones((d_nd.numel() // (N * N), N), bool),
                                               self.device).reshape(d_nd.shape)
                d_ninv = None
            else:
                d_ninv = hpx.to_dev(torch, ninv_diag, f64, self.device)
                if self.per_time and tuple(d_ninv.shape) == (nbl, N):
                    d_ninv = d_ninv[:, None, :].expand(nbl, T, N).contiguous()
                if tuple(d_ninv.shape) != ((nbl, T, N) if self.per_time else (nbl, N)):
                    raise ValueError(f"inverse noise variances of shape {tuple(d_ninv.shape)} do not match flags of shape "
                                     f"{tuple(flags.shape)}: expected (Nbl, Nfreqs) or, time dependent, (Nbl, Ntimes, Nfreqs)")
            fg_shared = len(fg_shape) == 2
            d_fg = hpx.to_dev(torch, fgmodes, c128, self.device)
            pmap, xgrid = _prior_tables(ps_prior if isinstance(ps_prior, np.ndarray)
                                        else ps_prior.detach().cpu().numpy(), N)
            prior_shared = pmap.ndim == 1
            d_pmap = hpx.to_dev(torch, pmap, torch.int32, self.device)
            d_xgrid = hpx.to_dev(torch, xgrid, f64, self.device) if len(xgrid) else None
            if tables is None:
                tables = draw_tables(T, N, self.Niter, seed, reseed=not map_estimate)
            uni, igy = tables
            assert uni.shape == (self.Niter, N) and igy.shape == (self.Niter, N)
            if omega is None:
                omega = omega_table(T, N)
            assert tuple(omega.shape) == (T, 4, N)
            d_omega = None if map_estimate else hpx.to_dev(torch, omega, f64, self.device)
            d_fop = hpx.to_dev(torch, utils.fourier_operator(N), c128, self.device)
            self.plan = hpx.Plan(nbl, T, N, M, extra_rhs=extra_rhs)
            L = hpx.lib()
            if not allow_split:
                hpx.set_option(hpx.OPT_FACTOR_SPLIT, 0, plan=self.plan)
            if self.per_time and self.dense_noise:
                hpx.check(L.hpx_plan_set_static_pertime_dense(
                    self.plan.handle, hpx.ptr(d_vis), hpx.ptr(d_flags), hpx.ptr(d_nd), hpx.ptr(d_nh),
                    hpx.ptr(d_fg) if M > 0 else None, int(fg_shared), hpx.ptr(d_pmap), hpx.ptr(d_xgrid),
                    int(len(xgrid)), int(prior_shared), NGRID, hpx.ptr(d_omega), hpx.ptr(d_fop),
                    int(self.any_flags), hpx.stream_ptr(torch)), "hpx_plan_set_static_pertime_dense")
            elif self.per_time:
                # every time sample has its own flags / noise, hence its own system: Nbl x Ntimes
                # factorisations per iteration (the mode the reference documents but does not
                # implement: pspec.py:337-340, :398-401, FIXMEs :361, :450-451)
                hpx.check(L.hpx_plan_set_static_pertime(
                    self.plan.handle, hpx.ptr(d_vis), hpx.ptr(d_flags), hpx.ptr(d_ninv),
                    hpx.ptr(d_fg) if M > 0 else None, int(fg_shared), hpx.ptr(d_pmap), hpx.ptr(d_xgrid),
                    int(len(xgrid)), int(prior_shared), NGRID, hpx.ptr(d_omega), hpx.ptr(d_fop),
                    int(self.any_flags), hpx.stream_ptr(torch)), "hpx_plan_set_static_pertime")
            elif self.dense_noise and self.any_flags:
                # non-Hermitian in the reference (pspec.py:361 FIXME): a rank-f update of the unflagged-noise
                # system, solved through the Woodbury identity (hpx.h)
                hpx.check(L.hpx_plan_set_static_dense_flagged(
                    self.plan.handle, hpx.ptr(d_vis), hpx.ptr(d_flags), hpx.ptr(d_nd), int(nd.ndim == 2),
                    hpx.ptr(d_nh), hpx.ptr(d_fg) if M > 0 else None, int(fg_shared), hpx.ptr(d_pmap),
                    hpx.ptr(d_xgrid), int(len(xgrid)), int(prior_shared), NGRID, hpx.ptr(d_omega), hpx.ptr(d_fop),
                    hpx.stream_ptr(torch)), "hpx_plan_set_static_dense_flagged")
            elif self.dense_noise:
                hpx.check(L.hpx_plan_set_static_dense(
                    self.plan.handle, hpx.ptr(d_vis), hpx.ptr(d_flags), hpx.ptr(d_nd), hpx.ptr(d_nh),
                    int(nd.ndim == 2), hpx.ptr(d_fg) if M > 0 else None, int(fg_shared), hpx.ptr(d_pmap),
                    hpx.ptr(d_xgrid), int(len(xgrid)), int(prior_shared), NGRID, hpx.ptr(d_omega), hpx.ptr(d_fop),
                    0, hpx.stream_ptr(torch)), "hpx_plan_set_static_dense")
            else:
                hpx.check(L.hpx_plan_set_static(
                    self.plan.handle, hpx.ptr(d_vis), hpx.ptr(d_flags), hpx.ptr(d_ninv),
                    hpx.ptr(d_fg) if M > 0 else None, int(fg_shared), hpx.ptr(d_pmap), hpx.ptr(d_xgrid),
                    int(len(xgrid)), int(prior_shared), NGRID, hpx.ptr(d_omega), hpx.ptr(d_fop),
                    int(self.any_flags), hpx.stream_ptr(torch)), "hpx_plan_set_static")
            self.set_tables(uni, igy)
            # "auto": baselines whose unflagged channels share one noise variance take a structured
            # solve (diagonal + border system): hpx_flat.hip without flags, hpx_lowrank.hip with
            # flags (border widened by one column per flagged channel); everything else the dense
            # Cholesky.  "flat" / "lowrank" insist on the structured solve, "dense" forbids it.
            assert solver in ("auto", "dense", "flat", "lowrank", "lowrank-direct"), \
                "solver must be auto, dense, flat, lowrank or lowrank-direct"
            # "lowrank-direct": the low-rank solve with the border laid out explicitly (MFMA form) even
            # where the FFT form applies (power-of-two Nfreqs, Nmodes <= 16); same results, for A/B
            direct = solver == "lowrank-direct"
            if direct:
                solver = "lowrank"
            self.solver = "dense"
            if self.dense_noise or self.per_time:
                if solver not in ("auto", "dense"):
                    raise ValueError("a non-diagonal inverse noise covariance / time-dependent flags need "
                                     "solver='dense' (or 'auto')")
                solver = "dense"
            if solver != "dense":
                # The library decides whether a structured solver applies (one inverse noise variance
                # over the unflagged channels of every baseline, size limits: hpx_plan_set_solver
                # inspects the plan's own copies once); "auto" falls back to the dense factorisation,
                # an explicit request that does not apply is an error.
                cand = "lowrank" if self.any_flags else "flat"
                want = cand if solver == "auto" else solver
                mode = hpx.SOLVER_FLAT if want == "flat" else \
                    (hpx.SOLVER_LOWRANK_DIRECT if direct else hpx.SOLVER_LOWRANK)
                size_ok = T <= 256 and (M <= 16 and N <= 4096 if want == "flat" else True)
                rc = L.hpx_plan_set_solver(self.plan.handle, mode) if (want == cand and size_ok) else hpx.HPX_EINVAL
                if rc == hpx.HPX_OK:
                    self.solver = want
                elif rc == hpx.HPX_EINVAL and solver == "auto":
                    self.solver = "dense"
                elif rc == hpx.HPX_EINVAL:
                    raise ValueError(f"solver={solver!r} does not apply: it needs one inverse noise variance over the "
                                     "unflagged channels of every baseline, Ntimes <= 256 and "
                                     + ("no flags, Nmodes <= 16" if solver == "flat"
                                        else "flags with Nmodes + max flagged channels <= 240")
                                     + (f" ({hpx.last_error()})" if want == cand and size_ok else ""))
                else:
                    hpx.check(rc, "hpx_plan_set_solver")
        self.iter_done = 0

    def close(self):
        self.plan.close()

    def set_tables(self, uni, igy):
        """(Re)load the bandpower draw's random tables, each (Niter, Nfreqs): uniforms and
        ``1/gammainccinv(Ntimes-1, U)`` (:func:`draw_tables`).  Host arrays are uploaded on the
        current stream; the call returns when the plan owns a copy."""
        torch = self.torch
        assert tuple(uni.shape) == (self.Niter, self.N) and tuple(igy.shape) == (self.Niter, self.N)
        with torch.cuda.device(self.device):
            d_uni = hpx.to_dev(torch, uni, torch.float64, self.device)
            d_igy = hpx.to_dev(torch, igy, torch.float64, self.device)
            hpx.check(hpx.lib().hpx_plan_set_rng(self.plan.handle, hpx.ptr(d_uni), hpx.ptr(d_igy),
                                                 self.Niter, hpx.stream_ptr(torch)), "hpx_plan_set_rng")

    def run(self, niter, ps0=None, ps_forced=None, keep=("ps", "ln_post"), thin=1, shp0=None):
        """Advance every chain by ``niter`` iterations.

        ``shp0`` (nbl,N,N) complex, first call only: general starting covariance given as
        ``Sh' = U^H sqrtm(S_initial) U`` (:func:`sqrt_cov_delay_basis`) instead of ``ps0``; the
        first iteration then runs through ``hpx_gibbs_step_general``.

        ps0 (nbl,N): bandpowers of the starting covariance (required on the
        first call).  Returns a dict of device tensors: ``signal_ps``
        (nbl,niter,N), ``ln_post`` (nbl,niter), ``ps_last`` (nbl,N) and, if named
        in ``keep``, ``signal_cr`` (nbl,nkeep,T,N) c128, ``fg_amps``
        (nbl,nkeep,T,M) c128, ``chisq`` (nbl,nkeep,T,N)."""
        torch = self.torch
        nbl, T, N, M = self.nbl, self.T, self.N, self.M
        assert self.iter_done + niter <= self.Niter, "random tables exhausted"
        if shp0 is not None:
            return self._run_general_first(niter, shp0, ps_forced, keep, thin)
        assert ps0 is not None or self.iter_done > 0, "ps0 is required for the first run"
        nkeep = (niter + thin - 1) // thin
        with torch.cuda.device(self.device):
            f64, c128 = torch.float64, torch.complex128
            dev = self.device
            d_ps0 = None if ps0 is None else hpx.to_dev(torch, ps0, f64, dev)
            if d_ps0 is not None:
                assert tuple(d_ps0.shape) == (nbl, N)
            d_forced = None if ps_forced is None else hpx.to_dev(torch, ps_forced, f64, dev)
            if d_forced is not None:
                assert tuple(d_forced.shape) == (nbl, niter, N)
            out = dict(signal_ps=torch.empty((nbl, niter, N), dtype=f64, device=dev),
                       ln_post=torch.empty((nbl, niter), dtype=f64, device=dev),
                       ps_last=torch.empty((nbl, N), dtype=f64, device=dev))
            if "signal_cr" in keep:
                out["signal_cr"] = torch.empty((nbl, nkeep, T, N), dtype=c128, device=dev)
            if "fg_amps" in keep:
                out["fg_amps"] = torch.zeros((nbl, nkeep, T, M), dtype=c128, device=dev)
            if "chisq" in keep:
                out["chisq"] = torch.empty((nbl, nkeep, T, N), dtype=f64, device=dev)
            rc = hpx.lib().hpx_gibbs_run(
                self.plan.handle, hpx.ptr(d_ps0), self.iter_done, niter, hpx.ptr(d_forced),
                hpx.ptr(out["signal_ps"]), hpx.ptr(out["ln_post"]), hpx.ptr(out.get("signal_cr")),
                hpx.ptr(out.get("fg_amps")), hpx.ptr(out.get("chisq")), thin,
                hpx.ptr(out["ps_last"]), hpx.stream_ptr(torch))
            hpx.check(rc, "hpx_gibbs_run")
        self.iter_done += niter
        return out

    def _run_general_first(self, niter, shp0, ps_forced, keep, thin):
        torch = self.torch
        nbl, T, N, M = self.nbl, self.T, self.N, self.M
        assert self.iter_done == 0, "a general starting covariance only makes sense for iteration 0"
        with torch.cuda.device(self.device):
            f64, c128, dev = torch.float64, torch.complex128, self.device
            d_shp = hpx.to_dev(torch, shp0, c128, dev)
            assert tuple(d_shp.shape) == (nbl, N, N)
            first = dict(signal_ps=torch.empty((nbl, 1, N), dtype=f64, device=dev),
                         ln_post=torch.empty((nbl, 1), dtype=f64, device=dev),
                         ps_last=torch.empty((nbl, N), dtype=f64, device=dev))
            if "signal_cr" in keep:
                first["signal_cr"] = torch.empty((nbl, 1, T, N), dtype=c128, device=dev)
            if "fg_amps" in keep:
                first["fg_amps"] = torch.zeros((nbl, 1, T, M), dtype=c128, device=dev)
            if "chisq" in keep:
                first["chisq"] = torch.empty((nbl, 1, T, N), dtype=f64, device=dev)
            rc = hpx.lib().hpx_gibbs_step_general(
                self.plan.handle, hpx.ptr(d_shp), 0, hpx.ptr(first["signal_ps"]), hpx.ptr(first["ln_post"]),
                hpx.ptr(first.get("signal_cr")), hpx.ptr(first.get("fg_amps")), hpx.ptr(first.get("chisq")),
                hpx.ptr(first["ps_last"]), hpx.stream_ptr(torch))
            hpx.check(rc, "hpx_gibbs_step_general")
            self.iter_done = 1
            if ps_forced is not None:     # teacher forcing: iteration 1 starts from the forced value
                forced = hpx.to_dev(torch, ps_forced, f64, dev)
                nxt_ps0 = forced[:, 0].contiguous()
            else:
                nxt_ps0 = None
            if niter == 1:
                return first
            rest = self.run(niter - 1, ps0=nxt_ps0,
                            ps_forced=None if ps_forced is None else forced[:, 1:].contiguous(),
                            keep=keep, thin=1)
            out = {}
            for k in first:
                out[k] = rest[k] if k == "ps_last" else torch.cat([first[k], rest[k]], dim=1)
            if thin > 1:
                for k in ("signal_cr", "fg_amps", "chisq"):
                    if k in out:
                        out[k] = out[k][:, ::thin].contiguous()
            return out


def make_batch(vis, flags, fgmodes, Ninv, ps_prior, Niter, seed=None, map_estimate=False, device=None,
               solver="auto", tables=None, allow_split=True):
    """A :class:`GibbsBatch` from the inverse noise covariance in any form the path accepts:
    diagonals ``(Nfreqs,)`` / ``(Nbl,Nfreqs)`` or matrices ``(Nfreqs,Nfreqs)`` / ``(Nbl,Nfreqs,Nfreqs)``
    (reference run-hydra-pspec.py:427-438 passes ``inv(noise_cov)``)."""
    nbl, T, N = tuple(vis.shape)
    # (a 2-D (Nbl, Nfreqs) array with Nbl > 1 is a stack of diagonals, also when Nbl == Nfreqs; a matrix shared
    # by Nbl == Nfreqs baselines has to be given as (Nbl, Nfreqs, Nfreqs))
    per_time_ninv = np.shape(Ninv) in ((T, N, N), (nbl, T, N, N), (nbl, T, N)) and (len(np.shape(Ninv)) == 4 or T != nbl
                                                                                    or np.shape(Ninv) == (nbl, T, N))
    diag_stack = tuple(np.shape(Ninv)) == (nbl, N) and nbl > 1
    nd = _ninv_dense(Ninv, nbl, N) if len(np.shape(Ninv)) >= 2 and not per_time_ninv and not diag_stack else None
    if nd is None and per_time_ninv and len(np.shape(Ninv)) >= 3:
        nd = _ninv_dense_pertime(Ninv, nbl, T, N)        # None when the matrices are diagonal
    if nd is not None and nd.ndim == 3 and len(tuple(flags.shape)) == 3:
        nd = np.ascontiguousarray(np.broadcast_to(nd[:, None], (nbl, T, N, N)))    # time-dependent flags only
    elif nd is not None and nd.ndim == 2 and len(tuple(flags.shape)) == 3:
        nd = np.ascontiguousarray(np.broadcast_to(nd, (nbl, T, N, N)))
    if nd is not None:
        return GibbsBatch(vis, flags, fgmodes, None, ps_prior, Niter, seed=seed, map_estimate=map_estimate,
                          device=device, solver=solver, ninv_dense=nd, tables=tables, allow_split=allow_split)
    return GibbsBatch(vis, flags, fgmodes, _ninv_diag(Ninv, nbl, T, N), ps_prior, Niter, seed=seed,
                      map_estimate=map_estimate, device=device, solver=solver, tables=tables, allow_split=allow_split)


def gibbs_sample_with_fg_batched(vis, flags, fgmodes, Ninv, ps_prior, S_initial=None,
                                 ps_initial=None, Niter=100, seed=None, map_estimate=False,
                                 keep=("ps", "ln_post"), thin=1, ps_forced=None, device=None,
                                 as_numpy=True, iter0=0, solver="auto", allow_split=True):
    """Run the Gibbs chain of ``gibbs_sample_with_fg`` for ``Nbl`` baselines at once.

    Parameters mirror the reference (pspec.py:493-571) with a leading baseline
    axis: ``vis`` (Nbl,Ntimes,Nfreqs) complex, ``flags`` (Nbl,Nfreqs) bool
    (True = use), ``fgmodes`` (Nfreqs,Nmodes) or (Nbl,Nfreqs,Nmodes), ``Ninv``
    diagonal (Nbl,Nfreqs)/(Nfreqs,) or matrices (Nfreqs,Nfreqs)/(Nbl,Nfreqs,Nfreqs) --
    diagonal ones, or Hermitian with off-diagonal terms (a correlated noise covariance;
    dense solver; with flags through a Woodbury correction) --, ``ps_prior`` (2,Nfreqs) or (Nbl,2,Nfreqs) with
    rows [hi, lo].  Time-dependent flags ``(Nbl,Ntimes,Nfreqs)``, optionally with per-time
    inverse noise variances ``Ninv`` (Nbl,Ntimes,Nfreqs) or matrices (Ntimes,Nfreqs,Nfreqs) /
    (Nbl,Ntimes,Nfreqs,Nfreqs) (diagonal, or Hermitian with off-diagonal terms), select the mode in which every time sample is solved with its own
    noise matrix (Nbl x Ntimes factorisations per iteration; memory: one factor buffer of
    (Nfreqs+Nmodes)^2 x 16 bytes, ~ 5 MB at Nfreqs = 512, per baseline and time).  The initial covariance is
    given either as ``ps_initial`` (Nbl,Nfreqs)/(Nfreqs,) bandpowers
    (``S = F^H diag(ps/N^2) F``) or as ``S_initial`` matrices of that form.
    ``seed`` is shared by all baselines, as in the reference driver.

    Returns a dict: ``signal_ps`` (Nbl,Niter,Nfreqs), ``ln_post`` (Nbl,Niter),
    ``ps_last`` and the histories named in ``keep`` (``"signal_cr"``,
    ``"fg_amps"``, ``"chisq"``; every ``thin``-th iteration).

    ``solver``: ``"auto"`` (default) solves baselines whose unflagged channels share one ``Ninv``
    value through the diagonal + border structure of the system (hpx_flat.hip without flags,
    hpx_lowrank.hip with flags) and everything else with the batched dense Cholesky; ``"dense"``
    forces the latter.  All are exact solves.  ``allow_split=False``: see :class:`GibbsBatch` (several
    processes on one GPU; bit-identical chains at every batch size).

    ``iter0 > 0`` continues interrupted chains: ``ps_initial`` must then be the bandpowers of
    iteration ``iter0 - 1`` and only iterations ``iter0 .. Niter-1`` are run and returned (the
    random tables are generated for all ``Niter`` iterations, so the samples are those of an
    uninterrupted run)."""
    nbl, T, N = tuple(vis.shape)
    shp0 = None
    if ps_initial is None:
        if S_initial is None:
            raise ValueError("one of S_initial / ps_initial is required")
        S0 = np.asarray(S_initial)
        ps_initial, resid = pspec_from_covariance(S0)
        if np.any(resid > FOURIER_FORM_TOL):     # general covariance: first iteration via Sh'
            shp0 = np.ascontiguousarray(np.broadcast_to(sqrt_cov_delay_basis(S0), (nbl, N, N)))
    gb = make_batch(vis, flags, fgmodes, Ninv, ps_prior, Niter, seed=seed, map_estimate=map_estimate,
                    device=device, solver=solver, allow_split=allow_split)
    try:
        if shp0 is not None:
            assert iter0 == 0, "a general S_initial cannot be combined with iter0 > 0"
            out = gb.run(gb.Niter, shp0=shp0, ps_forced=ps_forced, keep=keep, thin=thin)
        else:
            ps0 = np.ascontiguousarray(np.broadcast_to(np.asarray(ps_initial, dtype=float), (nbl, N)))
            assert 0 <= iter0 < gb.Niter
            gb.iter_done = iter0
            out = gb.run(gb.Niter - iter0, ps0=ps0, ps_forced=ps_forced, keep=keep, thin=thin)
    finally:
        gb.close()
    if as_numpy:
        out = {k: v.cpu().numpy() for k, v in out.items()}
    return out


# ------------------------------------------------------- reference call surface
class GcrMatrices(list):
    """Return type of :func:`build_matrices`: indexes like the reference's two-element list
    (``m[0]`` = (4,N,N) ``Sh, S, Ni, Nih``; ``m[1]`` = (2,n,n) ``A, Ai``) and also carries the
    inputs the device solver works from."""
    flags = None
    signal_S = None
    ninv_diag = None
    fgmodes = None


def _gpu_herm_solve(K, B):
    """X = K^-1 B for Hermitian positive definite K (n,n) and B (n,m), on the GPU
    (batched Cholesky + triangular solves of hpx_factor.hip)."""
    torch = hpx.require_gpu()
    n, m = K.shape[0], B.shape[1]
    dev = torch.device("cuda", torch.cuda.current_device())
    dK = hpx.to_dev(torch, np.asarray(K, dtype=complex)[None], torch.complex128, dev)
    dB = hpx.to_dev(torch, np.asarray(B, dtype=complex)[None], torch.complex128, dev)
    dX = torch.empty((1, n, m), dtype=torch.complex128, device=dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    hpx.check(hpx.lib().hpx_zpotrs_batched(1, n, m, hpx.ptr(dK), hpx.ptr(dB), hpx.ptr(dX), hpx.ptr(info),
                                           hpx.stream_ptr(torch)), "hpx_zpotrs_batched")
    if int(info.item()) != 0:
        raise np.linalg.LinAlgError("GCR system is not positive definite")
    return dX[0].cpu().numpy()


def build_matrices(Nparams, flags, signal_S, Ninv, fgmodes):
    """The operators of the GCR system in the reference's layout (pspec.py:325-374):
    ``[ (4,N,N) = Sh, S, Ni, Nih ; (2,Nparams,Nparams) = A, Ai ]``.

    The sampler itself never forms these (it factorises the equivalent Hermitian system,
    DESIGN.md section 2); this function exists for callers that inspect or reuse them.  ``Sh``
    is the principal square root via a Hermitian eigendecomposition, ``Ni = Ninv * flags``
    (column mask, as the reference) and ``Nih`` its element-wise root for the diagonal ``Ninv``
    the path supports.  ``Ai`` -- the reference's ``pinv(A)`` preconditioner -- is the exact
    inverse ``P K'^-1 P^-1`` with ``P = diag(Sh, I)`` and ``K'`` the Hermitian positive-definite
    matrix the device factorises; it therefore needs a positive-definite ``signal_S``."""
    S = np.asarray(signal_S, dtype=complex)
    N = S.shape[0]
    fl = np.asarray(flags)
    F = np.asarray(fgmodes, dtype=complex)
    M = F.shape[1]
    assert Nparams == N + M, "Nparams must equal Nfreqs + Nmodes"
    dense = _ninv_dense(Ninv, 1, N)
    ops = np.zeros((4, N, N), dtype=complex)
    lam, V = np.linalg.eigh(0.5 * (S + S.conj().T))
    if lam.min() <= 1e-14 * lam.max():
        raise np.linalg.LinAlgError("build_matrices needs a positive-definite signal_S")
    ops[0] = (V * np.sqrt(lam)) @ V.conj().T
    ops[1] = S
    if dense is not None:
        # Hermitian non-diagonal Ninv: the reference's operators as it forms them (pspec.py:359-372, host
        # linear algebra: this function is not on the sampler's path) -- Ni = Ninv with its flagged COLUMNS
        # zeroed, Nih = scipy's sqrtm of that matrix, A the non-Hermitian system, Ai = pinv(A)
        import scipy.linalg
        nd = np.asarray(dense, dtype=complex).reshape(N, N)
        w = fl.reshape(-1).astype(bool)
        ops[2] = nd * w[None, :]
        ops[3] = sqrtm_hermitian(nd) if w.all() else scipy.linalg.sqrtm(ops[2])
        sys_ = np.zeros((2, Nparams, Nparams), dtype=complex)
        A = sys_[0]
        A[:N, :N] = np.eye(N) + S @ ops[2]
        A[:N, N:] = S @ ops[2] @ F
        A[N:, :N] = F.conj().T @ ops[2]
        A[N:, N:] = F.conj().T @ ops[2] @ F
        sys_[1] = np.linalg.pinv(A)
        out = GcrMatrices([ops, sys_])
        out.flags, out.signal_S, out.ninv_diag, out.fgmodes = fl, S, None, F
        return out
    ninv = _ninv_diag(Ninv, 1, 1, N)[0]
    ni = ninv * fl
    ops[2] = np.diag(ni)
    ops[3] = np.diag(np.sqrt(ni))
    sys_ = np.zeros((2, Nparams, Nparams), dtype=complex)
    A = sys_[0]
    A[:N, :N] = np.eye(N) + S * ni[None, :]
    A[:N, N:] = (S * ni[None, :]) @ F
    A[N:, :N] = F.conj().T * ni[None, :]
    A[N:, N:] = (F.conj().T * ni[None, :]) @ F
    # Ai = P K'^-1 P^-1,  K' = [[I + Sh Ni Sh, Sh Ni F], [F^H Ni Sh, F^H Ni F]]
    Sh = ops[0]
    Kp = np.empty_like(A)
    Kp[:N, :N] = np.eye(N) + (Sh * ni[None, :]) @ Sh
    Kp[:N, N:] = (Sh * ni[None, :]) @ F
    Kp[N:, :N] = Kp[:N, N:].conj().T
    Kp[N:, N:] = A[N:, N:]
    Pinv = np.eye(Nparams, dtype=complex)
    Pinv[:N, :N] = (V / np.sqrt(lam)) @ V.conj().T
    X = _gpu_herm_solve(Kp, Pinv)
    X[:N] = Sh @ X[:N]
    sys_[1] = X
    out = GcrMatrices([ops, sys_])
    out.flags, out.signal_S, out.ninv_diag, out.fgmodes = fl, S, ninv, F
    return out


def _hermitian_completion(Ni, fl):
    """A Hermitian positive-definite ``H`` with ``H diag(fl) = Ni`` for the reference's column-masked
    ``Ni = Ninv diag(fl)`` (pspec.py:361), when only ``Ni`` is at hand (the ``matrices`` argument of the GCR entry
    points): the unflagged columns are Ni's, the flagged rows of those columns give the flagged-by-unflagged block and
    its adjoint, and the flagged-by-flagged block -- which the system never sees -- is chosen to keep ``H`` positive
    definite (Schur complement = the mean unflagged diagonal times the identity)."""
    Ni = np.asarray(Ni, dtype=complex)
    u = np.asarray(fl, dtype=bool)
    A = Ni[np.ix_(u, u)]
    if np.abs(A - A.conj().T).max() > 1e-12 * np.abs(A).max():
        raise NotImplementedError("a non-Hermitian inverse noise covariance is not supported")
    if u.all():
        return 0.5 * (Ni + Ni.conj().T)
    H = np.zeros_like(Ni)
    B = Ni[np.ix_(~u, u)]
    H[np.ix_(u, u)] = 0.5 * (A + A.conj().T)
    H[np.ix_(~u, u)] = B
    H[np.ix_(u, ~u)] = B.conj().T
    C = B @ np.linalg.solve(H[np.ix_(u, u)], B.conj().T)
    H[np.ix_(~u, ~u)] = 0.5 * (C + C.conj().T) + np.eye(int((~u).sum())) * np.real(np.diagonal(A)).mean()
    return H


def _gcr_solve(vis2d, w, matrices, fgmodes, map_estimate, idx, seed0=GCR_SEED0):
    """Constrained realisations [s_t ; f_t] for the rows of ``vis2d`` on the GPU, with the
    reference's per-time noise streams (seeded ``seed0 + t``) for time indices ``idx``."""
    vis2d = np.asarray(vis2d, dtype=complex)
    if vis2d.shape[0] == 1:      # a plan holds at least two times (the draw's shape is Ntimes - 1)
        idx = [0] if idx is None else idx
        return _gcr_solve(np.repeat(vis2d, 2, axis=0), w, matrices, fgmodes, map_estimate,
                          [idx[0], idx[0]], seed0=seed0)[:1]
    T, N = vis2d.shape
    F = np.asarray(fgmodes, dtype=complex)
    M = F.shape[1]
    S = np.asarray(matrices[0][1])
    Ni = np.asarray(matrices[0][2])
    ni = np.real(np.diagonal(Ni))          # already column-masked
    fl = np.asarray(w).reshape(-1).astype(bool)
    ps0, resid = pspec_from_covariance(S)
    tables = (np.full((1, N), 0.5), np.ones((1, N)))     # the bandpower draw's output is discarded
    if np.abs(Ni - np.diag(np.diagonal(Ni))).max() > 0:
        # a non-diagonal Ni: the chain's dense-noise path (GibbsBatch(ninv_dense=...): Hermitian noise factorisation
        # + one Woodbury column per flagged channel) on a Hermitian matrix whose unflagged columns are Ni's
        gb = GibbsBatch(vis2d[None], fl[None], F, None, np.zeros((2, N)), 1, map_estimate=map_estimate, tables=tables,
                        omega=omega_table(T, N, idx=idx, seed0=seed0), ninv_dense=_hermitian_completion(Ni, fl)[None],
                        solver="dense")
    else:
        gb = GibbsBatch(vis2d[None], fl[None], F, np.ascontiguousarray(ni)[None], np.zeros((2, N)), 1,
                        map_estimate=map_estimate, tables=tables, omega=omega_table(T, N, idx=idx, seed0=seed0))
    try:
        if resid > FOURIER_FORM_TOL:
            out = gb.run(1, shp0=sqrt_cov_delay_basis(S)[None], keep=("signal_cr", "fg_amps"))
        else:
            out = gb.run(1, ps0=ps0[None], keep=("signal_cr", "fg_amps"))
    finally:
        gb.close()
    return np.concatenate([out["signal_cr"][0, 0].cpu().numpy(), out["fg_amps"][0, 0].cpu().numpy()], axis=1)


def gcr_fgmodes_1d(idx, vis, w, matrices, fgmodes, f0=None, map_estimate=False, verbose=False,
                   multiprocess_seed=GCR_SEED0):
    """GCR step for one time sample (reference pspec.py:151-235): returns
    ``(xsoln (Nfreqs+Nmodes,), residual, info)``.  The noise realisation is the reference's
    (stream seeded with ``multiprocess_seed + idx``, any seed: pspec.py:196-197), and like the reference the call
    leaves numpy's GLOBAL stream seeded with that value and advanced past the four draws; the system is solved
    directly on the GPU, so ``f0`` is accepted and ignored and ``info`` is always 0.  ``residual`` (verbose only) is
    the mean absolute residual of the reference's own system ``A x = b``."""
    F = np.asarray(fgmodes)
    N = F.shape[0]
    d = np.asarray(vis).reshape(1, N)
    seed0 = int(multiprocess_seed)
    x = _gcr_solve(d, w, matrices, fgmodes, map_estimate, [idx], seed0=seed0)[0]
    np.random.seed(seed0 + int(idx))           # the reference's side effect on the global stream (pspec.py:196-216)
    if not map_estimate:
        np.random.randn(4, N)
    residual = None
    if verbose:
        Sh, S, Ni, Nih, A = matrices[0][0], matrices[0][1], matrices[0][2], matrices[0][3], matrices[1][0]
        o = np.zeros((4, N)) if map_estimate else omega_table(1, N, idx=[idx], seed0=seed0)[0]
        oma, omb = (o[0] + 1j * o[1]) / 2 ** 0.5, (o[2] + 1j * o[3]) / 2 ** 0.5
        wd = (np.asarray(w).reshape(-1) * d[0])
        b = np.concatenate([S @ (Ni @ wd) + Sh @ oma + S @ (Nih @ omb),
                            F.conj().T @ (Ni @ wd + Nih @ omb)])
        residual = np.abs(A @ x - b).mean()
    return x, residual, 0


def gcr_fgmodes(vis, w, matrices, fgmodes, f0=None, nproc=1, map_estimate=False, verbose=False):
    """GCR step for all time samples of one baseline (reference pspec.py:238-310): returns
    samples of shape ``(Ntimes, Nfreqs + Nmodes)`` = rows ``[s_t ; f_t]``.  All times are
    solved in one batched GPU call; ``nproc`` and ``f0`` are accepted and ignored."""
    vis = np.asarray(vis)
    return _gcr_solve(vis, w, matrices, fgmodes, map_estimate, None)


def covariance_from_pspec(ps, fourier_op):
    """``fourier_op^H diag(ps) fourier_op`` (reference pspec.py:313-322), computed
    with the batched DFT kernel: row j of ``(F * ps)`` is transformed by F^H."""
    torch = hpx.require_gpu()
    ps = np.asarray(ps, dtype=float)
    N = ps.size
    dev = torch.device("cuda", torch.cuda.current_device())
    fop = hpx.to_dev(torch, np.asarray(fourier_op, dtype=complex), torch.complex128, dev)
    rows = (fop * hpx.to_dev(torch, ps, torch.float64, dev)[None, :]).contiguous()   # F[j,k] ps[k]
    # the operator is symmetric: F^T = F, so column j of diag(ps) F is row j of F diag(ps)
    out = torch.empty((1, N, N), dtype=torch.complex128, device=dev)
    hpx.check(hpx.lib().hpx_dft_batched(1, N, N, hpx.ptr(fop), hpx.ptr(rows), hpx.ptr(out), 1,
                                        hpx.stream_ptr(torch)), "hpx_dft_batched")
    return (out[0].T * N).cpu().numpy()


def inversion_sample_invgamma(alpha, beta, prior_min, prior_max, ngrid=1000):
    """Truncated inverse-gamma draw by inversion of the CDF sampled on a log grid
    (reference pspec.py:11-64).  Consumes one ``np.random.uniform()``.  The HIP
    kernel evaluates the CDF for integer ``alpha`` (the path always passes
    ``alpha = Ntimes``)."""
    if prior_min <= 0:
        raise ValueError("prior_min must be greater than zero")
    if prior_max <= 0:
        raise ValueError("prior_max must be greater than zero")
    if not np.isfinite(prior_max):
        raise ValueError("prior_max must be finite")
    if prior_max <= prior_min:
        raise ValueError("prior_max must be greater than prior_min")
    if alpha <= 0:
        raise ValueError("alpha must be greater than zero")
    if float(alpha) != int(alpha):
        return _inversion_sample_invgamma_host(float(alpha), float(beta), prior_min, prior_max, ngrid)
    torch = hpx.require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device())
    x = np.logspace(np.log10(prior_min), np.log10(prior_max), ngrid)
    u = np.random.uniform()
    f64 = torch.float64
    d_b = hpx.to_dev(torch, np.array([beta], dtype=float), f64, dev)
    d_u = hpx.to_dev(torch, np.array([u]), f64, dev)
    d_x = hpx.to_dev(torch, x[None, :], f64, dev)
    d_o = torch.empty(1, dtype=f64, device=dev)
    hpx.check(hpx.lib().hpx_invgamma_inversion(1, int(alpha), hpx.ptr(d_b), hpx.ptr(d_u),
                                               hpx.ptr(d_x), ngrid, hpx.ptr(d_o),
                                               hpx.stream_ptr(torch)), "hpx_invgamma_inversion")
    return float(d_o.cpu()[0])


def _inversion_sample_invgamma_host(alpha, beta, prior_min, prior_max, ngrid):
    """The same draw for a NON-INTEGER shape parameter, evaluated on the host (reference pspec.py:49-62 step
    by step: the inverse-gamma CDF Q(alpha, beta / x) on the log grid, shifted and rescaled to [0, 1],
    duplicate values dropped, linear interpolation of the grid at one ``np.random.uniform()``).  The device kernel
    sums the closed form of Q for integer alpha (all the Gibbs path ever passes: alpha = Ntimes); a general alpha
    needs the incomplete gamma function itself, a scalar utility call outside the per-baseline path."""
    from scipy.special import gammaincc
    x = np.logspace(np.log10(prior_min), np.log10(prior_max), ngrid)
    cdf = gammaincc(alpha, beta / x)
    cdf = cdf - cdf.min()
    cdf = cdf / cdf.max()
    cdf_unique, idx = np.unique(cdf, return_index=True)
    u = np.random.uniform()
    return float(np.interp(u, cdf_unique, x[idx]))


def sample_S(s=None, sk=None, prior=None):
    """Bandpower draw p(S|s) (reference pspec.py:67-127): one global uniform per
    channel; ``x = beta * invgamma.ppf(U, Nobs-1)`` or the truncated draw with
    shape ``Nobs`` inside the prior box."""
    if s is None and sk is None:
        raise ValueError("Must pass in s (real space) or sk (Fourier space) vector.")
    torch = hpx.require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device())
    c128, f64 = torch.complex128, torch.float64
    if sk is None:
        s = np.asarray(s, dtype=complex)
        nobs, nfreq = s.shape
        d_s = hpx.to_dev(torch, s[None], c128, dev)
        d_sk = torch.empty_like(d_s)
        d_fop = hpx.to_dev(torch, utils.fourier_operator(nfreq), c128, dev)
        hpx.check(hpx.lib().hpx_dft_batched(1, nobs, nfreq, hpx.ptr(d_fop), hpx.ptr(d_s), hpx.ptr(d_sk),
                                            0, hpx.stream_ptr(torch)), "hpx_dft_batched")
        sk_t = d_sk[0]
    else:
        sk = np.asarray(sk, dtype=complex)
        nobs, nfreq = sk.shape
        sk_t = hpx.to_dev(torch, sk, c128, dev)
    d_beta = torch.empty(nfreq, dtype=f64, device=dev)
    hpx.check(hpx.lib().hpx_power_sum(1, nobs, nfreq, hpx.ptr(sk_t.contiguous()), hpx.ptr(d_beta),
                                      hpx.stream_ptr(torch)), "hpx_power_sum")
    beta = d_beta.cpu().numpy()
    if prior is None:
        prior = np.zeros((2, nfreq))
    pmap, xgrid = _prior_tables(prior, nfreq)
    u = np.random.random_sample(nfreq)
    x = beta / scipy.special.gammainccinv(nobs - 1.0, u)
    sel = np.nonzero(pmap >= 0)[0]
    if sel.size:
        d_b = hpx.to_dev(torch, beta[sel], f64, dev)
        d_u = hpx.to_dev(torch, u[sel], f64, dev)
        d_x = hpx.to_dev(torch, xgrid[pmap[sel]], f64, dev)
        d_o = torch.empty(sel.size, dtype=f64, device=dev)
        hpx.check(hpx.lib().hpx_invgamma_inversion(int(sel.size), int(nobs), hpx.ptr(d_b), hpx.ptr(d_u),
                                                   hpx.ptr(d_x), NGRID, hpx.ptr(d_o),
                                                   hpx.stream_ptr(torch)), "hpx_invgamma_inversion")
        x[sel] = d_o.cpu().numpy()
    return x


def sprior(signals, bins, factor):
    """Prior box derived from data (reference pspec.py:130-148; unused by the
    chain).  ``|fft(v)|^2`` is the centred transform's power, un-shifted."""
    torch = hpx.require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device())
    signals = np.asarray(signals, dtype=complex)
    nobs, nfreq = signals.shape
    c128 = torch.complex128
    d_s = hpx.to_dev(torch, np.fft.fftshift(signals, axes=-1)[None], c128, dev)
    d_sk = torch.empty_like(d_s)
    d_fop = hpx.to_dev(torch, utils.fourier_operator(nfreq), c128, dev)
    hpx.check(hpx.lib().hpx_dft_batched(1, nobs, nfreq, hpx.ptr(d_fop), hpx.ptr(d_s), hpx.ptr(d_sk), 0,
                                        hpx.stream_ptr(torch)), "hpx_dft_batched")
    d_ds = torch.empty(nfreq, dtype=torch.float64, device=dev)
    hpx.check(hpx.lib().hpx_power_sum(1, nobs, nfreq, hpx.ptr(d_sk), hpx.ptr(d_ds), hpx.stream_ptr(torch)),
              "hpx_power_sum")
    ds = np.fft.ifftshift(d_ds.cpu().numpy())
    prior = np.zeros((2, nfreq))
    prior[0] = ds * factor
    prior[1] = ds / factor
    prior[0, bins + 1:-bins] = 0
    prior[1, bins + 1:-bins] = 0
    return prior / (nobs / 2 - 1)


def gibbs_step_fgmodes(vis, flags, signal_S, fgmodes, Ninv, ps_prior=None, f0=None, nproc=1,
                       map_estimate=False, verbose=False):
    """One Gibbs iteration (reference pspec.py:377-490).  Draws its N uniforms
    from numpy's global stream without reseeding, like the reference.  ``f0`` and
    ``nproc`` are accepted and ignored (a direct solve needs no initial guess)."""
    vis = np.asarray(vis)
    N = vis.shape[1]
    assert flags.shape == (N,), "`flags` array must have shape (Nfreqs,)"
    if ps_prior is None:
        ps_prior = np.zeros((2, N))
    ps0, resid = pspec_from_covariance(np.asarray(signal_S))
    T = vis.shape[0]
    gb = make_batch(vis[None], np.asarray(flags)[None], fgmodes, Ninv, ps_prior, 1,
                    map_estimate=map_estimate, tables=draw_tables(T, N, 1, None, reseed=False))
    try:
        if resid > FOURIER_FORM_TOL:
            out = gb.run(1, shp0=sqrt_cov_delay_basis(np.asarray(signal_S))[None],
                         keep=("signal_cr", "fg_amps", "chisq"))
        else:
            out = gb.run(1, ps0=ps0[None], keep=("signal_cr", "fg_amps", "chisq"))
    finally:
        gb.close()
    ps_sample = out["signal_ps"][0, 0].cpu().numpy()
    S_sample = covariance_from_pspec(ps_sample / N ** 2, utils.fourier_operator(N))
    return (out["signal_cr"][0, 0].cpu().numpy(), S_sample, ps_sample,
            out["fg_amps"][0, 0].cpu().numpy(), out["chisq"][0, 0].cpu().numpy(),
            float(out["ln_post"][0, 0].cpu()))


def gibbs_sample_with_fg(vis, flags, S_initial, fgmodes, Ninv, ps_prior, Niter=100, seed=None,
                         verbose=True, nproc=1, write_Niter=100, out_dir=None, map_estimate=False,
                         resume=False, solver="auto"):
    """Drop-in for the reference chain driver (pspec.py:493-658).

    Returns ``(signal_cr (Niter,Ntimes,Nfreqs) c128, signal_S (Nfreqs,Nfreqs)
    c128 [last sample only, as in the reference], signal_ps (Niter,Nfreqs),
    fg_amps (Niter,Ntimes,Nmodes) c128, chisq (Niter,Ntimes,Nfreqs), ln_post
    (Niter,), write_time)``.  ``nproc`` is accepted and ignored: the reference's
    results do not depend on it.  ``map_estimate=True`` forces ``Niter = 1`` and
    does not reseed (pspec.py:572-577).

    ``resume=True`` (extension; the reference writes checkpoints but cannot read them back,
    pspec.py:625-636): if ``out_dir`` holds the six sample files of an interrupted run of
    the SAME inputs and seed with ``k < Niter`` iterations, the chain continues from
    iteration ``k`` and produces exactly the samples an uninterrupted run would have."""
    vis = np.asarray(vis)
    flags = np.asarray(flags)
    if map_estimate:
        Niter = 1
        write_Niter = 1
    Ntimes, Nfreqs = vis.shape
    Nmodes = fgmodes.shape[1]
    assert flags.shape in ((Nfreqs,), (Ntimes, Nfreqs)), \
        "`flags` array must have shape (Nfreqs,) or, time dependent (extension), (Ntimes, Nfreqs)"
    assert fgmodes.shape[0] == Nfreqs, "fgmodes must have shape (Nfreqs, Nmodes)"
    if len(np.shape(Ninv)) == 3:
        assert np.shape(Ninv)[0] == Ntimes, \
            "Ninv shape must be (Ntimes, Nfreqs, Nfreqs) or (Nfreqs, Nfreqs)"
    ps0, resid = pspec_from_covariance(np.asarray(S_initial))
    shp0 = sqrt_cov_delay_basis(np.asarray(S_initial))[None] if resid > FOURIER_FORM_TOL else None
    fop = utils.fourier_operator(Nfreqs)
    gb = make_batch(vis[None], flags[None], fgmodes, Ninv, ps_prior, Niter, seed=seed,
                    map_estimate=map_estimate, solver=solver)
    signal_cr = np.zeros((Niter, Ntimes, Nfreqs), dtype=complex)
    signal_ps = np.zeros((Niter, Nfreqs))
    fg_amps = np.zeros((Niter, Ntimes, Nmodes), dtype=complex)
    chisq = np.zeros((Niter, Ntimes, Nfreqs))
    ln_post = np.zeros(Niter)
    signal_S = np.asarray(S_initial).copy()
    if verbose:
        print("Iter     Time [s]    Chisq    ln Post")
        print("-----    --------    -----    -------")
    write_time = 0
    done = 0
    chunk = max(1, int(write_Niter)) if out_dir is not None else Niter
    if resume and out_dir is not None and not map_estimate:
        from pathlib import Path
        od = Path(out_dir)
        if all((od / f).exists() for f in utils.SAMPLE_FILES):
            prev = [np.load(od / f) for f in utils.SAMPLE_FILES]
            k = prev[2].shape[0]
            if 0 < k < Niter and prev[0].shape == (k, Ntimes, Nfreqs) and prev[3].shape == (k, Ntimes, Nmodes):
                signal_cr[:k], signal_ps[:k], fg_amps[:k], chisq[:k], ln_post[:k] = \
                    prev[0], prev[2], prev[3], prev[4], prev[5]
                gb.iter_done = done = k
                ps0, shp0 = signal_ps[k - 1].copy(), None
    resumed_at = done
    try:
        while done < Niter:
            n = min(chunk - done % chunk, Niter - done)
            t0 = time.perf_counter()
            if done == 0 and shp0 is not None:
                out = gb.run(n, shp0=shp0, keep=("signal_cr", "fg_amps", "chisq"))
            else:
                out = gb.run(n, ps0=ps0[None] if done == resumed_at else None,
                             keep=("signal_cr", "fg_amps", "chisq"))
            sl = slice(done, done + n)
            signal_cr[sl] = out["signal_cr"][0].cpu().numpy()
            signal_ps[sl] = out["signal_ps"][0].cpu().numpy()
            fg_amps[sl] = out["fg_amps"][0].cpu().numpy()
            chisq[sl] = out["chisq"][0].cpu().numpy()
            ln_post[sl] = out["ln_post"][0].cpu().numpy()
            done += n
            signal_S = covariance_from_pspec(signal_ps[done - 1] / Nfreqs ** 2, fop)
            if verbose:
                dt = (time.perf_counter() - t0) / n
                for i in range(done - n, done):
                    cm = chisq[i][:, flags].mean() if flags.ndim == 1 else chisq[i][flags].mean()
                    print(f"{i + 1:<9d}{dt:<12.3g}{cm:<9.3f}{ln_post[i]:<12.1f}")
            if out_dir is not None and done % write_Niter == 0:
                # periodic checkpoint: everything so far; cov-eor.npy gets rows [:done] of
                # the CURRENT covariance, as in the reference (pspec.py:625-636)
                tw = time.perf_counter()
                utils.write_numpy_files(out_dir, signal_cr[:done], signal_S[:done], signal_ps[:done],
                                        fg_amps[:done], chisq[:done], ln_post[:done])
                write_time += time.perf_counter() - tw
    finally:
        gb.close()
    if out_dir is not None and Niter % write_Niter > 0:
        tw = time.perf_counter()
        utils.write_numpy_files(out_dir, signal_cr, signal_S, signal_ps, fg_amps, chisq, ln_post)
        write_time += time.perf_counter() - tw
    if verbose:
        print()
    return signal_cr, signal_S, signal_ps, fg_amps, chisq, ln_post, write_time
