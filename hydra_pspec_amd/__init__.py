"""MI355X-native implementation of hydra-pspec's per-baseline Gibbs inner loop.

Mirrors the reference package layout for the hot path only
(reference hydra_pspec/__init__.py:12): ``pspec`` (sampler), ``utils``
(``fourier_operator``, ``write_numpy_files``), ``dpss`` and ``oqe``.  All
arithmetic of the sampler runs in hand-written HIP kernels (``csrc/``) behind
the C-ABI declared in ``include/hpx.h``; there is no CPU fallback -- calling a
compute entry point without the built library or without a GPU raises.
"""
__version__ = "0.1.0"

from . import utils  # noqa: F401  (pure host helpers, no GPU needed)


def __getattr__(name):
    # pspec/dpss/oqe bind the HIP library; import them lazily so that host-only
    # tools (synthetic data, file writers) work on a machine without a GPU.
    if name in ("pspec", "dpss", "oqe", "synthetic", "hpx", "fgmodes", "uvh5", "h5lite"):
        import importlib
        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
