"""ctypes binding of libhpx.so (include/hpx.h) on torch ROCm tensors.

torch is used for device memory and streams only; every number the sampler
produces comes out of the HIP kernels behind this ABI.  There is no fallback:
a missing library or a missing GPU raises.
"""
import ctypes as C
import os
from pathlib import Path

import numpy as np

_LIB = None
# HPX_LIB_PATH: another build of the same library (A/B measurements of kernel variants, tools/build_variant.sh)
_LIB_PATH = Path(os.environ.get("HPX_LIB_PATH") or Path(__file__).resolve().parent / "libhpx.so")

HPX_OK, HPX_EINVAL, HPX_EHIP, HPX_ENOTPD, HPX_ETIMEOUT = 0, -1, -2, -3, -4
OPT_FACTOR_SPLIT, OPT_SPLIT_HEAVY, OPT_SPLIT_SPIN_LIMIT, OPT_EIGH_INNER_SWEEPS, OPT_EIGH_TRACE = 1, 2, 3, 4, 5
OPT_SPLIT_RETRY, OPT_SPLIT_FALLBACKS = 6, 7
INFO_TIMEOUT = 0x40000000
NSTAGE = 6
SOLVER_DENSE, SOLVER_FLAT, SOLVER_LOWRANK, SOLVER_LOWRANK_DIRECT = 0, 1, 2, 3
STAGES = ("assemble", "factor", "backsolve", "transform", "residual", "draw")

_vp, _i, _i64 = C.c_void_p, C.c_int, C.c_int64

# name -> (restype, argtypes); must list every symbol declared in include/hpx.h
SIGNATURES = {
    "hpx_version": (_i, []),
    "hpx_last_error": (C.c_char_p, []),
    "hpx_device_count": (_i, []),
    "hpx_set_device": (_i, [_i]),
    "hpx_plan_create": (_i, [C.POINTER(_vp), _i, _i, _i, _i]),
    "hpx_plan_create_ex": (_i, [C.POINTER(_vp), _i, _i, _i, _i, _i]),
    "hpx_plan_destroy": (_i, [_vp]),
    "hpx_plan_bytes": (_i64, [_vp]),
    "hpx_plan_set_static": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    "hpx_plan_set_static_dense": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    "hpx_plan_set_static_dense_flagged": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "hpx_plan_set_static_pertime": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    "hpx_plan_set_static_pertime_dense": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    "hpx_plan_set_rng": (_i, [_vp, _vp, _vp, _i, _vp]),
    "hpx_gibbs_run": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "hpx_gibbs_step_general": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hpx_plan_info": (_i, [_vp, _vp]),
    "hpx_set_option": (_i, [_vp, _i, _i]),
    "hpx_get_option": (_i, [_vp, _i, _vp]),
    "hpx_plan_set_profiling": (_i, [_vp, _i]),
    "hpx_plan_set_solver": (_i, [_vp, _i]),
    "hpx_plan_stage_ms": (_i, [_vp, _vp]),
    "hpx_assemble_K": (_i, [_vp, _vp, _vp, _vp]),
    "hpx_plan_dims": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "hpx_factor_form": (_i, [_i, _i, _i, C.POINTER(_i)]),
    "hpx_zpotrf_batched": (_i, [_i, _i, _vp, _vp, _vp, _vp]),
    "hpx_zpotrs_batched": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "hpx_dft_batched": (_i, [_i, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "hpx_invgamma_inversion": (_i, [_i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "hpx_dpss_project": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hpx_dpss_workspace_bytes": (_i64, [_i, _i, _i, _i]),
    "hpx_dpss_group_info": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _vp]),
    "hpx_dpss_project_grouped": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _vp]),
    "hpx_oqe_workspace_bytes": (_i64, [_i, _i, _i]),
    "hpx_oqe_fisher": (_i, [_i, _i, _vp, _vp, _i, _vp, _i64, _vp]),
    "hpx_oqe_qh": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _i64, _vp]),
    "hpx_oqe_sandwich_diag": (_i, [_i, _i, _vp, _vp, _i, _vp, _vp, _i64, _vp]),
    "hpx_oqe_mopt": (_i, [_i, _i, _vp, _vp, _vp]),
    "hpx_power_sum": (_i, [_i, _i, _i, _vp, _vp, _vp]),
    "hpx_lincomb": (_i, [_i64, C.c_double, _vp, C.c_double, _vp, _vp, _vp]),
    "hpx_fgmodes_eig": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "hpx_oqe_qauto": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _i64, _vp]),
    "hpx_zheev_psd_batched": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp]),
    "hpx_zheev_psd_order": (_i, [_i]),
    "hpx_sqrtm_hpd_batched": (_i, [_i, _i, _vp, _vp, _vp, C.c_double, _i, _vp, _vp]),
    "hpx_sqrtm_masked_batched": (_i, [_i, _i, _vp, _i, _vp, _vp, C.c_double, _i, _vp, _vp]),
    "hpx_mfma_probe": (_i, [_vp, _vp, _vp]),
    "hpx_mfma_f64_peak": (_i, [_i, _vp]),
}


def lib():
    """Load libhpx.so (built by ``__graft_entry__.build()`` / csrc/Makefile)."""
    global _LIB
    if _LIB is None:
        if not _LIB_PATH.exists():
            raise RuntimeError(
                f"{_LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C hydra_pspec_amd/csrc). "
                "hydra_pspec_amd has no CPU fallback.")
        # torch ships its own libamdhip64 (same soname as /opt/rocm's).  Import it first so
        # that libhpx binds to the HIP runtime torch already loaded: one runtime per process,
        # otherwise torch's streams and device pointers would belong to a different runtime.
        import torch  # noqa: F401
        L = C.CDLL(str(_LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        _LIB = L
    return _LIB


def last_error():
    return lib().hpx_last_error().decode(errors="replace")


class HpxTimeout(RuntimeError):
    """HPX_ETIMEOUT: a hand-off between the workgroups of a split factorisation timed out -- the GPU is shared with
    another process running the same form (set OPT_FACTOR_SPLIT to 0), not a property of the data.  hpx_gibbs_run
    repeats such a run once without the form (OPT_SPLIT_RETRY) before it would report this."""


def check(rc, what=""):
    if rc == HPX_OK:
        return
    msg = f"{what}: {last_error()}" if what else last_error()
    if rc == HPX_EINVAL:
        raise ValueError(msg)
    if rc == HPX_ENOTPD:
        raise FloatingPointError(msg)
    if rc == HPX_ETIMEOUT:
        raise HpxTimeout(msg)
    raise RuntimeError(msg)


def set_option(key, value, plan=None):
    """hpx_set_option: library-wide (plan=None) or for one plan (OPT_* keys)."""
    check(lib().hpx_set_option(plan.handle if plan is not None else None, int(key), int(value)), "hpx_set_option")


def get_option(key, plan):
    """hpx_get_option: the current value of a plan option (OPT_FACTOR_SPLIT, OPT_SPLIT_RETRY, OPT_SPLIT_FALLBACKS)."""
    v = C.c_int(0)
    check(lib().hpx_get_option(plan.handle, int(key), C.byref(v)), "hpx_get_option")
    return v.value


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("hydra_pspec_amd needs a ROCm GPU (MI355X); none is visible and there "
                           "is no CPU fallback")
    return torch


def ptr(t):
    """Device pointer of a contiguous torch tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "tensor must be contiguous"
    return C.c_void_p(t.data_ptr())


def stream_ptr(torch):
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def to_dev(torch, x, dtype, device):
    """numpy / torch (any device) -> contiguous torch tensor of `dtype` on `device`."""
    if isinstance(x, np.ndarray):
        x = np.ascontiguousarray(x)
        # torch refuses read-only arrays (broadcast views, arrays out of an .npz) and negative
        # strides (which survive ascontiguousarray in length-1 dimensions)
        if not x.flags.writeable or any(st < 0 for st in x.strides):
            x = x.copy()
        x = torch.from_numpy(x)
    return x.to(device=device, dtype=dtype, non_blocking=False).contiguous()


class Plan:
    """RAII wrapper of hpx_plan (one batch of baselines on one GPU)."""

    def __init__(self, nbl, T, N, M, extra_rhs=0):
        self._h = C.c_void_p()
        if extra_rhs:
            check(lib().hpx_plan_create_ex(C.byref(self._h), nbl, T, N, M, int(extra_rhs)), "hpx_plan_create_ex")
        else:
            check(lib().hpx_plan_create(C.byref(self._h), nbl, T, N, M), "hpx_plan_create")
        self.nbl, self.T, self.N, self.M = nbl, T, N, M
        self._keep = []

    def close(self):
        if getattr(self, "_h", None) is not None and self._h and lib is not None:
            lib().hpx_plan_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def handle(self):
        return self._h

    def bytes(self):
        return int(lib().hpx_plan_bytes(self._h))

    def dims(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        check(lib().hpx_plan_dims(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def stage_ms(self):
        buf = (C.c_float * NSTAGE)()
        check(lib().hpx_plan_stage_ms(self._h, buf))
        return dict(zip(STAGES, [float(v) for v in buf]))

    def set_profiling(self, on):
        check(lib().hpx_plan_set_profiling(self._h, int(bool(on))))

    def info(self):
        out = np.zeros(self.nbl, dtype=np.int32)
        check(lib().hpx_plan_info(self._h, out.ctypes.data_as(C.c_void_p)))
        return out
