"""DPSS fit and OQE helpers on the GPU against the reference's own outputs (golden small.npz)."""
import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu


def _cost(p, modes, d, w, cov, taper):
    mm = np.sum(p[0::2, None] * modes + 1j * p[1::2, None] * modes, axis=0)
    x = (1.0 if taper is None else taper) * w * (d - mm)
    return (0.5 * np.dot(x.conj(), np.linalg.inv(cov) @ x)).real


def test_dpss_fit_modes_vs_reference(golden):
    from hydra_pspec_amd import dpss
    from oracle import dpss_ref
    g = golden("small")
    for i in range(3):
        nm, al, has_t = g[f"F10_{i}_par"]
        taper = g[f"F10_{i}_taper"] if has_t else None
        d, w, fr, cov = g[f"F10_{i}_d"], g[f"F10_{i}_w"], g[f"F10_{i}_freqs"], g[f"F10_{i}_cov"]
        modes, amps = dpss.dpss_fit_modes(d, w, fr, cov, nmodes=int(nm), alpha=al, taper=taper)
        assert np.array_equal(modes, g[f"F10_{i}_modes"])
        scale = np.max(np.abs(g[f"F10_{i}_amps"]))
        # reference = L-BFGS-B stop point: within its slack, and never a better cost than ours
        assert np.max(np.abs(amps - g[f"F10_{i}_amps"])) < 1e-3 * scale
        assert _cost(amps, modes, d, w, cov, taper) <= _cost(g[f"F10_{i}_amps"], modes, d, w, cov, taper) * (1 + 1e-12)
        _, cf = dpss_ref.dpss_fit_closed_form(d, w, fr, cov, nmodes=int(nm), alpha=al, taper=taper)
        assert np.max(np.abs(amps - cf)) < 1e-9 * scale       # same minimiser as the CPU closed form


def test_dpss_batched_matches_single_and_complex_cov():
    from hydra_pspec_amd import dpss
    from oracle import dpss_ref
    rng = np.random.default_rng(3)
    N, nm, nb = 96, 8, 5
    freqs = np.linspace(100., 120., N)
    a = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    cov = a @ a.conj().T / N + np.eye(N)                      # Hermitian, complex
    d = rng.standard_normal((nb, N)) + 1j * rng.standard_normal((nb, N))
    w = (rng.uniform(size=(nb, N)) > 0.2).astype(float)
    modes, amps = dpss.dpss_fit_modes_batched(d, w, freqs, cov, nmodes=nm, alpha=4.0)
    for b in range(nb):
        _, cf = dpss_ref.dpss_fit_closed_form(d[b], w[b], freqs, cov, nmodes=nm, alpha=4.0)
        assert np.max(np.abs(amps[b] - cf)) < 1e-9 * np.max(np.abs(cf))
    with pytest.raises(AssertionError):
        dpss.dpss_fit_modes(d[0], w[0, :-1], freqs, cov, nmodes=nm)


@pytest.mark.parametrize("s", [8, 16])
def test_oqe_vs_reference(golden, s):
    from hydra_pspec_amd import oqe
    g = golden("small")
    R, Rg, V, Cn = g[f"F11_{s}_R"], g[f"F11_{s}_Rg"], g[f"F11_{s}_V"], g[f"F11_{s}_Cn"]
    assert relerr(oqe.Q(3, s), g[f"F11_{s}_Q3"]) < 1e-15
    Fm = oqe.F(s, R)
    assert relerr(Fm, g[f"F11_{s}_F"]) < 1e-12
    assert relerr(oqe.Ft(s, R), g[f"F11_{s}_Ft"]) < 1e-12
    assert relerr(oqe.F(s, Rg), g[f"F11_{s}_Fg"]) < 1e-12        # non-Hermitian weighting
    assert relerr(oqe.Ft(s, Rg), g[f"F11_{s}_Ftg"]) < 1e-12
    assert relerr(oqe.F(s, np.stack([R, Rg])), np.stack([g[f"F11_{s}_F"], g[f"F11_{s}_Fg"]])) < 1e-12
    assert relerr(oqe.M_opt(Fm), g[f"F11_{s}_Mopt"]) < 1e-11
    assert relerr(oqe.M_Finv(Fm), g[f"F11_{s}_MFinv"]) < 1e-9
    assert relerr(oqe.q_h(V, s, R), g[f"F11_{s}_qh"]) < 1e-12
    assert relerr(oqe.q_h(V, s, Rg), g[f"F11_{s}_qhg"]) < 1e-12
    assert oqe.qhat_h(V[0], V[1], 2, s, R) == pytest.approx(g[f"F11_{s}_qh"][0, 2], rel=1e-12)
    qs, F2, MB, MA = oqe.getqs(V, R)
    assert relerr(qs, g[f"F11_{s}_qh"]) < 1e-12 and relerr(F2, g[f"F11_{s}_F"]) < 1e-12
    b = np.array([oqe.bias(t, s, R, Cn) for t in range(s)])
    assert relerr(b, g[f"F11_{s}_bias"]) < 1e-12
    assert relerr(oqe.q(V, s, R, b.real), g[f"F11_{s}_q"]) < 1e-12          # the reference's own output
    # auto-estimator with a general (non-Hermitian, non-symmetric) weighting too: the device form is
    # 1/2 conj(FFT(R^T x)) FFT(R x), the reference's is the explicit x^H (R^* Q_tau R) x
    for W in (R, Rg):
        ref = np.array([[0.5 * (x.conj() @ (W.conj() @ oqe.Q(t, s) @ W) @ x) - 0.25 * t for t in range(s)] for x in V])
        got = np.array([[oqe.qhat(x, t, s, W, 0.25 * t) for t in range(s)] for x in V[:2]])
        assert relerr(got, ref[:2]) < 1e-12
        assert relerr(oqe.q(V, s, W, 0.25 * np.arange(s)), ref.real) < 1e-12
    assert relerr(oqe.Sig_QEN(R, Cn, 0.37), g[f"F11_{s}_SigN"]) < 1e-12
    assert relerr(oqe.Sig_QESN(R, Cn, R, 0.37), g[f"F11_{s}_SigSN"]) < 1e-12


def test_sample_S_and_covariance_api(golden):
    """sample_S (pspec.py:67-127), covariance_from_pspec (:313-322), sprior (:130-148)."""
    from hydra_pspec_amd import pspec, utils
    g = golden("small")
    np.random.seed(11)
    assert relerr(pspec.sample_S(s=g["F4_s"]), g["F4_x_noprior"]) < 1e-10
    np.random.seed(11)
    assert relerr(pspec.sample_S(s=g["F4_s"], prior=g["F4_prior"]), g["F4_x_prior"]) < 1e-10
    assert np.random.uniform() == g["F4_next_uniform"]
    with pytest.raises(ValueError):
        pspec.sample_S()
    assert relerr(pspec.covariance_from_pspec(g["F2_ps"], utils.fourier_operator(8)), g["F2_cov"]) < 1e-13
    assert relerr(pspec.sprior(g["F4_s"], 2, 10.0), g["F5_prior"]) < 1e-12
