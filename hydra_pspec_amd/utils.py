"""Host-side helpers of the Gibbs path (mirror of ``hydra_pspec.utils`` for the
two functions the path uses; reference: hydra_pspec/utils.py:15-41, 272-312)."""
from pathlib import Path

import numpy as np

# File names of the six per-baseline sample arrays (reference utils.py:307-312).
SAMPLE_FILES = ("gcr-eor.npy", "cov-eor.npy", "dps-eor.npy", "fg-amps.npy",
                "chisq.npy", "ln-post.npy")


def fourier_operator(n):
    """Centred DFT matrix ``F[k, x] = exp(-2 pi i (k - n//2)(x - n//2) / n)``.

    ``F @ v == fftshift(fft(ifftshift(v)))`` (reference utils.py:15-41).  The
    product ``k*x/n`` is formed first and the phase factor applied afterwards,
    which reproduces the reference matrix bit for bit."""
    c = np.arange(n) - n // 2
    return np.exp(-2 * np.pi * 1j * (c[:, None] * c[None, :] / n))


def write_numpy_files(fp, signal_cr, signal_S, signal_ps, fg_amps, chisq, ln_post):
    """Save the six sample arrays under directory ``fp`` with the reference's
    file names (reference utils.py:272-312)."""
    fp = Path(fp)
    for name, arr in zip(SAMPLE_FILES, (signal_cr, signal_S, signal_ps, fg_amps,
                                        chisq, ln_post)):
        np.save(fp / name, arr)
