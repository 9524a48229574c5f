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
        assert np.max(np.abs(amps - g[f"F10_{i}_amps"])) < 1e-4 * scale       # (measured 1.4e-6 .. 1.9e-5)
        assert _cost(amps, modes, d, w, cov, taper) <= _cost(g[f"F10_{i}_amps"], modes, d, w, cov, taper) * (1 + 1e-12)
        _, cf = dpss_ref.dpss_fit_closed_form(d, w, fr, cov, nmodes=int(nm), alpha=al, taper=taper)
        assert np.max(np.abs(amps - cf)) < 1e-9 * scale       # same minimiser as the CPU closed form
        # SURVEY 8(a) D1's gate against the reference with its optimiser allowed to converge (dpss_control.npz)
        tight = golden("dpss_control")[f"F10_{i}_amps_tight"]
        assert np.max(np.abs(amps - tight)) < 1e-6 * scale


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


@pytest.mark.parametrize("s", [8, 16])
def test_oqe_normalisations_on_device(golden, s):
    """M_Fhalf (Denman-Beavers on the batched Cholesky solver), M_Finv, M_opt against the reference's
    own outputs (oqe.py:69-84)."""
    from hydra_pspec_amd import oqe
    g = golden("small")
    Fm = g[f"F11_{s}_F"]
    assert relerr(oqe.M_Fhalf(Fm), g[f"F11_{s}_MFhalf"]) < 1e-9
    assert relerr(oqe.M_Finv(Fm), g[f"F11_{s}_MFinv"]) < 1e-9
    assert relerr(oqe.M_opt(Fm), g[f"F11_{s}_Mopt"]) < 1e-11
    # a non-Hermitian matrix takes the LAPACK fallback and still matches numpy
    A = Fm + 0.1 * np.triu(np.ones((s, s)), 1)
    assert relerr(oqe.M_Finv(A), np.linalg.inv(A)) < 1e-12


def test_oqe_at_bench_size_vs_closed_form():
    """s = 512 (the bench shape; the reference's O(s^5) loops cannot run there): the device Fisher
    matrices, q_h and the noise terms against the numpy closed forms of the oracle."""
    from hydra_pspec_amd import oqe
    from oracle import oqe_ref
    s, nb = 512, 2
    rng = np.random.default_rng(5)
    a = (rng.standard_normal((nb, s, s)) + 1j * rng.standard_normal((nb, s, s))) / np.sqrt(s)
    R = a @ a.conj().transpose(0, 2, 1) + np.eye(s)
    Fd = oqe.F(s, R)
    Ftd = oqe.Ft(s, R)
    for b in range(nb):
        assert relerr(Fd[b], oqe_ref.F_closed(s, R[b])) < 1e-11
        assert relerr(Ftd[b], oqe_ref.Ft_closed(s, R[b])) < 1e-11
    V = rng.standard_normal((6, s)) + 1j * rng.standard_normal((6, s))
    assert relerr(oqe.q_h(V, s, R[0]), oqe_ref.q_h_closed(V, s, R[0])) < 1e-11
    Cn = np.diag(rng.uniform(0.5, 1.5, s)).astype(complex)
    Mm = np.fft.fft(np.eye(s))
    ref_bias = 0.5 * np.diagonal(Mm @ (R[0] @ Cn @ R[0].conj()) @ Mm.conj().T)
    assert relerr(oqe.bias_all(s, R[0], Cn), ref_bias) < 1e-11
    n = np.diagonal(Mm @ (R[0] @ Cn @ R[0]) @ Mm.conj().T)
    assert relerr(oqe.Sig_QEN(R[0], Cn, 0.5), 0.5 * 0.25 * n * n) < 1e-11


def test_dpss_grouped_at_bench_size_vs_closed_form():
    """The (baseline x time) cube at the bench shape (N = 512, 12 modes, weights shared by the 32
    times of a baseline): a sample of spectra against the CPU closed form; groups of one spectrum
    (hpx_dpss_project) give the same amplitudes."""
    from hydra_pspec_amd import dpss
    from oracle import dpss_ref
    rng = np.random.default_rng(8)
    ng, per, N, nm = 6, 32, 512, 12
    freqs = np.linspace(100., 200., N)
    x = np.arange(N)
    cov = np.exp(-0.5 * ((x[:, None] - x[None, :]) / 3.0) ** 2) + 0.5 * np.eye(N)     # smooth + white
    d = rng.standard_normal((ng, per, N)) + 1j * rng.standard_normal((ng, per, N))
    w = (rng.uniform(size=(ng, N)) > 0.15).astype(float)
    modes, amps = dpss.dpss_fit_modes_batched(d, w, freqs, cov, nmodes=nm, alpha=6.0)
    assert amps.shape == (ng, per, 2 * nm)
    for gi, t in ((0, 0), (2, 17), (5, 31)):
        _, cf = dpss_ref.dpss_fit_closed_form(d[gi, t], w[gi], freqs, cov, nmodes=nm, alpha=6.0)
        assert np.max(np.abs(amps[gi, t] - cf)) < 1e-9 * np.max(np.abs(cf))
    _, single = dpss.dpss_fit_modes_batched(d[2, :3], np.broadcast_to(w[2], (3, N)).copy(), freqs, cov,
                                            nmodes=nm, alpha=6.0)
    assert np.max(np.abs(single - amps[2, :3])) < 1e-11 * np.max(np.abs(single))
    # more than 16 modes: two mode tiles
    _, a20 = dpss.dpss_fit_modes_batched(d[1, :5], w[1], freqs, cov, nmodes=20, alpha=11.0)
    for t in (0, 4):
        _, cf = dpss_ref.dpss_fit_closed_form(d[1, t], w[1], freqs, cov, nmodes=20, alpha=11.0)
        assert np.max(np.abs(a20[t] - cf)) < 1e-8 * np.max(np.abs(cf))


def test_dpss_singular_group_gives_zero_amplitudes_and_a_warning():
    """A fully flagged spectrum (all-zero weights) has a singular weighted normal matrix: the reference's
    L-BFGS fit from a zero start returns zeros (dpss.py:81-92); here the group is marked, its amplitudes are
    zero (not NaN) and the batched entry warns.  The other groups are unaffected."""
    from hydra_pspec_amd import dpss
    from oracle import dpss_ref
    rng = np.random.default_rng(5)
    N, nm, nb = 64, 6, 4
    freqs = np.linspace(100., 110., N)
    cov = np.eye(N) * 2.0
    d = rng.standard_normal((nb, N)) + 1j * rng.standard_normal((nb, N))
    w = np.ones((nb, N))
    w[2] = 0.0
    with pytest.warns(RuntimeWarning, match="not positive definite"):
        modes, amps = dpss.dpss_fit_modes_batched(d, w, freqs, cov, nmodes=nm, alpha=3.0)
    assert np.isfinite(amps).all() and not amps[2].any()
    for b in (0, 1, 3):
        _, cf = dpss_ref.dpss_fit_closed_form(d[b], w[b], freqs, cov, nmodes=nm, alpha=3.0)
        assert np.max(np.abs(amps[b] - cf)) < 1e-9 * np.max(np.abs(cf))
    pr = dpss.DpssProjector(nb, 1, freqs, cov, nmodes=nm, alpha=3.0)
    pr.fit(d[:, None, :], w)
    assert list(pr.singular_groups()) == [2]


def test_m_fhalf_on_an_ill_conditioned_fisher_matrix():
    """``M_Fhalf`` = inv(sqrtm(F)) through the Denman-Beavers iteration on the batched Cholesky solver: converges
    for well-conditioned F, and for an ill-conditioned one (cond 1e10, where the iteration is not stable) the result
    is still inv(sqrtm(F)) -- the convergence / residual guard falls back to scipy instead of returning a stagnated
    iterate."""
    from hydra_pspec_amd import oqe
    rng = np.random.default_rng(12)
    s = 48
    q, _ = np.linalg.qr(rng.standard_normal((s, s)) + 1j * rng.standard_normal((s, s)))
    for span in (2, 6, 10):
        lam = np.logspace(0, -span, s)
        F = (q * lam) @ q.conj().T
        F = 0.5 * (F + F.conj().T)
        want = (q / np.sqrt(lam)) @ q.conj().T
        got = oqe.M_Fhalf(F)
        assert np.abs(got - want).max() < 1e-6 * np.abs(want).max(), span
        assert np.abs(got @ F @ got - np.eye(s)).max() < 1e-6
