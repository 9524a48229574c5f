"""Pin the CPU oracle (oracle/) to vectors produced by the real reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import pspec_ref as R, dpss_ref, oqe_ref
from conftest import relerr, GOLDEN

TIGHT = 1e-12


def test_fourier_operator_bits(golden):
    g = golden("small")
    for n in (4, 5, 8):
        assert np.array_equal(R.fourier_operator(n), g[f"F1_fop_{n}"])


def test_covariance_from_pspec(golden):
    g = golden("small")
    assert relerr(R.covariance_from_pspec(g["F2_ps"], R.fourier_operator(8)), g["F2_cov"]) < 1e-15


def test_inversion_sample_invgamma(golden):
    for a, beta, lo, hi, seed, v, u_after in golden("small")["F3_cases"]:
        np.random.seed(int(seed))
        got = float(R.inversion_sample_invgamma(a, beta, lo, hi))
        assert got == pytest.approx(v, rel=1e-13)
        assert np.random.uniform() == u_after      # exactly one uniform consumed


@pytest.mark.parametrize("bad", [(0.0, 1.0), (1.0, 0.0), (1.0, np.inf), (2.0, 1.0), (-1.0, 2.0)])
def test_inversion_sample_invgamma_errors(bad):
    with pytest.raises(ValueError):
        R.inversion_sample_invgamma(5.0, 5.0, bad[0], bad[1])


def test_sample_S(golden):
    g = golden("small")
    np.random.seed(11)
    assert relerr(R.sample_S(s=g["F4_s"]), g["F4_x_noprior"]) < TIGHT
    np.random.seed(11)
    assert relerr(R.sample_S(s=g["F4_s"], prior=g["F4_prior"]), g["F4_x_prior"]) < TIGHT
    assert np.random.uniform() == g["F4_next_uniform"]
    with pytest.raises(ValueError):
        R.sample_S()


def test_sprior(golden):
    g = golden("small")
    assert relerr(R.sprior(g["F4_s"], 2, 10.0), g["F5_prior"]) < 1e-15


@pytest.mark.parametrize("tag", ["nf", "fl"])
def test_build_matrices_and_gcr(golden, tag):
    g = golden("small")
    fl, S, Ninv, F = g[f"F6_{tag}_flags"], g[f"F6_{tag}_S_initial"], g[f"F6_{tag}_Ninv"], g[f"F6_{tag}_fgmodes"]
    mats = R.build_matrices(19, fl, S, Ninv, F)
    assert relerr(mats[0], g[f"F6_{tag}_ops"]) < TIGHT
    assert relerr(mats[1][0], g[f"F6_{tag}_sys"][0]) < TIGHT
    assert relerr(mats[1][1], g[f"F6_{tag}_sys"][1]) < 1e-9     # pinv: SVD-conditioned
    vis = g[f"F6_{tag}_vis"] * fl
    for row, idx in enumerate((0, 3)):
        x, _, info = R.gcr_fgmodes_1d(idx, vis[idx], fl, mats, F)
        assert info == 0
        assert relerr(x, g[f"F7_{tag}_x"][row]) < 1e-10
    x, _, _ = R.gcr_fgmodes_1d(1, vis[1], fl, mats, F, map_estimate=True)
    assert relerr(x, g[f"F7_{tag}_xmap"]) < 1e-10


def test_omega_stream(golden):
    np.random.seed(R.GCR_SEED0 + 2)
    got = np.array([np.random.randn(6, 1)[:, 0] for _ in range(4)])
    assert np.array_equal(got, golden("small")["F7_omega_idx2_N6"])


def test_gcr_loop_preserves_parent_stream(golden):
    g = golden("small")
    fl, S, Ninv, F = g["F6_nf_flags"], g["F6_nf_S_initial"], g["F6_nf_Ninv"], g["F6_nf_fgmodes"]
    mats = R.build_matrices(19, fl, S, Ninv, F)
    np.random.seed(5)
    expect = np.random.uniform()
    np.random.seed(5)
    R.gcr_fgmodes(g["F6_nf_vis"], fl, mats, F)
    assert np.random.uniform() == expect


STEP_NAMES = list("abcdefghi")


@pytest.mark.parametrize("name", STEP_NAMES)
def test_gibbs_step(golden, name):
    g = golden("steps")
    vis, fl = g[f"{name}_in_vis"], g[f"{name}_in_flags"]
    np.random.seed(4242)
    cr, S_s, ps, fg, chi, lp = R.gibbs_step_fgmodes(
        vis * fl, fl, g[f"{name}_in_S"], g[f"{name}_in_fgmodes"], g[f"{name}_in_Ninv"], g[f"{name}_in_prior"])
    assert relerr(cr, g[f"{name}_cr"]) < 1e-9
    assert relerr(fg, g[f"{name}_fg"]) < 1e-9
    assert np.max(np.abs(ps / g[f"{name}_ps"] - 1)) < 1e-9
    assert relerr(S_s, g[f"{name}_S"]) < 1e-9
    assert relerr(chi, g[f"{name}_chisq"]) < 1e-8
    assert lp == pytest.approx(float(g[f"{name}_lnpost"]), rel=1e-9)


def test_map_estimate(golden):
    g = golden("steps")
    vis, fl = g["b_in_vis"], g["b_in_flags"]
    np.random.seed(3)
    res = R.gibbs_sample_with_fg(vis, fl, g["b_in_S"], g["b_in_fgmodes"], g["b_in_Ninv"], g["b_in_prior"],
                                 Niter=5, seed=99, map_estimate=True)
    assert res[0].shape[0] == 1                      # Niter forced to 1 (pspec.py:572-574)
    assert relerr(res[0], g["map_cr"]) < 1e-9
    assert np.max(np.abs(res[2] / g["map_ps"] - 1)) < 1e-9
    assert relerr(res[3], g["map_fg"]) < 1e-9


@pytest.mark.parametrize("b", [0, 2])
def test_chain_synth_prefix(golden, b):
    """First 12 iterations of the reference chain (same seed, same solver)."""
    g = golden("chain_synth")
    vis, fl = g[f"b{b}_vis"], g[f"b{b}_flags"]
    res = R.gibbs_sample_with_fg(vis, fl, g["S_initial"], g["fgmodes"], g["Ninv"], g["prior"],
                                 Niter=12, seed=int(g["seed"]))
    assert np.max(np.abs(res[2] / g[f"b{b}_ref_ps"][:12] - 1)) < 1e-8
    assert np.allclose(res[5], g[f"b{b}_ref_lnpost"][:12], rtol=1e-9)


def test_chain_synth_teacher_forced_direct(golden):
    """Exact-solve oracle, teacher-forced on the reference chain: every step
    within the reference's own CG noise (T1 protocol on the CPU)."""
    g = golden("chain_synth")
    ref = g["b1_ref_ps"]
    res = R.gibbs_sample_with_fg(g["b1_vis"], g["b1_flags"], g["S_initial"], g["fgmodes"], g["Ninv"],
                                 g["prior"], Niter=40, seed=int(g["seed"]), solver="direct", ps_forced=ref)
    assert np.max(np.abs(res[2] / ref[:40] - 1)) < 1e-6


def _dpss_cost(p, modes, d, w, cov, taper):
    """The reference objective, dpss.py:78-86."""
    m = np.sum(p[0::2, None] * modes + 1j * p[1::2, None] * modes, axis=0)
    x = (1.0 if taper is None else taper) * w * (d - m)
    return (0.5 * np.dot(x.conj(), np.linalg.inv(cov) @ x)).real


def test_dpss_fit(golden):
    g = golden("small")
    for i in range(3):
        nm, al, has_t = g[f"F10_{i}_par"]
        taper = g[f"F10_{i}_taper"] if has_t else None
        args = (g[f"F10_{i}_d"], g[f"F10_{i}_w"], g[f"F10_{i}_freqs"], g[f"F10_{i}_cov"])
        modes, amps = dpss_ref.dpss_fit_modes(*args, nmodes=int(nm), alpha=al, taper=taper)
        assert np.array_equal(modes, g[f"F10_{i}_modes"])
        scale = np.max(np.abs(g[f"F10_{i}_amps"]))
        assert np.max(np.abs(amps - g[f"F10_{i}_amps"])) < 1e-9 * scale
        _, amps_cf = dpss_ref.dpss_fit_closed_form(*args, nmodes=int(nm), alpha=al, taper=taper)
        # The reference stops where L-BFGS-B (finite-difference gradients, default
        # tolerances) stops; the closed form is the true minimiser of the same
        # quadratic: it must not be worse, and it sits within the optimiser's slack.
        assert np.max(np.abs(amps_cf - g[f"F10_{i}_amps"])) < 1e-4 * scale      # the optimiser's slack: measured 1.4e-6 .. 1.9e-5
        assert _dpss_cost(amps_cf, modes, *args[:2], args[3], taper) <= \
            _dpss_cost(g[f"F10_{i}_amps"], modes, *args[:2], args[3], taper) * (1 + 1e-12)
        # the CONTROL (tests/golden/dpss_control.npz): the reference itself with its optimiser allowed to converge
        # (ftol 1e-15, gtol 1e-12; make_golden.py gen_dpss_control) -- SURVEY 8(a) D1's gate, 1e-6 of max |c|
        # (measured 2.3e-7 .. 8.8e-7: what L-BFGS-B's finite-difference gradients leave)
        tight = golden("dpss_control")[f"F10_{i}_amps_tight"]
        assert np.max(np.abs(amps_cf - tight)) < 1e-6 * scale
        assert _dpss_cost(amps_cf, modes, *args[:2], args[3], taper) <= \
            _dpss_cost(tight, modes, *args[:2], args[3], taper) * (1 + 1e-12)


@pytest.mark.parametrize("s", [8, 16])
def test_oqe(golden, s):
    g = golden("small")
    Rm, Rg, V, Cn = g[f"F11_{s}_R"], g[f"F11_{s}_Rg"], g[f"F11_{s}_V"], g[f"F11_{s}_Cn"]
    assert relerr(oqe_ref.Q(3, s), g[f"F11_{s}_Q3"]) < 1e-15
    Fm = oqe_ref.F(s, Rm)
    assert relerr(Fm, g[f"F11_{s}_F"]) < TIGHT
    assert relerr(oqe_ref.Ft(s, Rm), g[f"F11_{s}_Ft"]) < TIGHT
    assert relerr(oqe_ref.F(s, Rg), g[f"F11_{s}_Fg"]) < TIGHT
    assert relerr(oqe_ref.Ft(s, Rg), g[f"F11_{s}_Ftg"]) < TIGHT
    assert relerr(oqe_ref.M_opt(Fm), g[f"F11_{s}_Mopt"]) < TIGHT
    assert relerr(oqe_ref.M_Finv(Fm), g[f"F11_{s}_MFinv"]) < 1e-10
    assert relerr(oqe_ref.M_Fhalf(Fm), g[f"F11_{s}_MFhalf"]) < 1e-9
    assert relerr(oqe_ref.q_h(V, s, Rm), g[f"F11_{s}_qh"]) < TIGHT
    assert relerr(oqe_ref.q_h(V, s, Rg), g[f"F11_{s}_qhg"]) < TIGHT
    b = np.array([oqe_ref.bias(t, s, Rm, Cn) for t in range(s)])
    assert relerr(b, g[f"F11_{s}_bias"]) < TIGHT
    with np.testing.suppress_warnings() as sup:
        sup.filter(np.exceptions.ComplexWarning)
        assert relerr(oqe_ref.q(V, s, Rm, b.real), g[f"F11_{s}_q"]) < TIGHT
    assert relerr(oqe_ref.Sig_QEN(Rm, Cn, 0.37), g[f"F11_{s}_SigN"]) < TIGHT
    assert relerr(oqe_ref.Sig_QESN(Rm, Cn, Rm, 0.37), g[f"F11_{s}_SigSN"]) < TIGHT


def test_chain_fullsize_c3_prefix(golden):
    """The oracle against the REFERENCE's own chain at BASELINE.json's channel count (C3 shape
    (32, 512, 12), unflagged): first two iterations, same seed, same CG solver."""
    g = golden("chain_fullsize")
    N = g["c3_vis"].shape[1]
    S0 = R.covariance_from_pspec(g["c3_ps0"] / N ** 2, R.fourier_operator(N))
    res = R.gibbs_sample_with_fg(g["c3_vis"], g["c3_flags"], S0, g["c3_fgmodes"], np.diag(g["c3_ninv_diag"]),
                                 g["c3_prior"], Niter=2, seed=int(g["c3_seed"]))
    assert np.max(np.abs(res[2] / g["c3_ps"][:2] - 1)) < 1e-7
    assert np.allclose(res[5], g["c3_lnpost"][:2], rtol=1e-8)


@pytest.mark.parametrize("tag,its", [("c3", (100, 199)), ("c3f", (60,)), ("c5f", (29,))])
def test_oracle_deep_in_the_long_reference_chains(golden, tag, its):
    """The oracle against the reference DEEP in its long chains (tests/golden/chain_long_<tag>.npz): iteration k is one
    oracle step from the reference's own bandpowers of iteration k - 1, with the global stream where the reference had
    it (seeded, then k * Nfreqs uniforms consumed: one per channel per iteration) and the same CG solver."""
    g = golden(f"chain_long_{tag}")
    N = g["vis"].shape[1]
    fop = R.fourier_operator(N)
    for k in its:
        S_prev = R.covariance_from_pspec(g["ref_ps"][k - 1] / N ** 2, fop)
        np.random.seed(int(g["seed"]))
        np.random.random_sample(k * N)
        o = R.gibbs_step_fgmodes(g["vis"] * g["flags"], g["flags"], S_prev, g["fgmodes"], np.diag(g["ninv_diag"]),
                                 g["prior"])
        ref = g["ref_ps"][k]
        live = ref > 1e-9 * np.median(ref)
        assert np.max(np.abs(o[2] / ref - 1)[live]) < 1e-6, (tag, k)
        # ln posterior: ~1e11 of cancellation with flags; and once a foreground-wedge channel has collapsed to ~1e-16
        # of the median (channels next to the prior window do, after ~100 iterations) its 1/ps term is chaotic
        if live.all():
            assert o[5] == pytest.approx(g["ref_lnpost"][k], rel=2e-5)
        if k == g["sel"][-1]:
            assert relerr(o[3], g["ref_fg_sel"][-1]) < 1e-6


def test_oqe_closed_forms_equal_the_loop_forms():
    """oracle/oqe_ref.py's closed forms (used by the GPU tests at s = 512) against its restatement of the
    reference's trace loops and against the reference's own outputs."""
    from oracle import oqe_ref
    g = dict(np.load(GOLDEN / "small.npz"))
    for s in (8, 16):
        for key, fkey, ftkey in (("R", "F", "Ft"), ("Rg", "Fg", "Ftg")):
            R = g[f"F11_{s}_{key}"]
            assert relerr(oqe_ref.F_closed(s, R), g[f"F11_{s}_{fkey}"]) < 1e-12
            assert relerr(oqe_ref.Ft_closed(s, R), g[f"F11_{s}_{ftkey}"]) < 1e-12
        V = g[f"F11_{s}_V"]
        assert relerr(oqe_ref.q_h_closed(V, s, g[f"F11_{s}_R"]), g[f"F11_{s}_qh"]) < 1e-12
        assert relerr(oqe_ref.q_h_closed(V, s, g[f"F11_{s}_Rg"]), g[f"F11_{s}_qhg"]) < 1e-12


def test_dense_noise_covariance_step_and_chain():
    """A banded Hermitian noise covariance without flags (golden steps_dense.npz = the reference's own
    output, run-hydra-pspec.py:427-438 / pspec.py:361-369): the oracle reproduces the reference."""
    g = dict(np.load(GOLDEN / "steps_dense.npz"))
    np.random.seed(4242)
    o = R.gibbs_step_fgmodes(g["in_vis"] * g["in_flags"], g["in_flags"], g["in_S"], g["in_fgmodes"], g["in_Ninv"],
                             g["in_prior"])
    assert np.max(np.abs(o[2] / g["step_ps"] - 1)) < 1e-7
    assert relerr(o[0], g["step_cr"]) < 1e-7 and relerr(o[3], g["step_fg"]) < 1e-7
    assert o[5] == pytest.approx(float(g["step_lnpost"]), rel=1e-6)
    r = R.gibbs_sample_with_fg(g["in_vis"], g["in_flags"], g["in_S"], g["in_fgmodes"], g["in_Ninv"], g["in_prior"],
                               Niter=6, seed=77)
    assert np.max(np.abs(r[2] / g["chain_ps"] - 1)) < 1e-6
    # ... and with flagged channels on top (column-masked, non-Hermitian Ni: pspec.py:361)
    fl = g["fl_flags"]
    np.random.seed(4242)
    o = R.gibbs_step_fgmodes(g["in_vis"] * fl, fl, g["in_S"], g["in_fgmodes"], g["in_Ninv"], g["in_prior"])
    assert np.max(np.abs(o[2] / g["fl_step_ps"] - 1)) < 1e-7
    assert relerr(o[0], g["fl_step_cr"]) < 1e-7 and relerr(o[3], g["fl_step_fg"]) < 1e-7
    assert o[5] == pytest.approx(float(g["fl_step_lnpost"]), rel=1e-6)
    r = R.gibbs_sample_with_fg(g["in_vis"], fl, g["in_S"], g["in_fgmodes"], g["in_Ninv"], g["in_prior"], Niter=6, seed=77)
    assert np.max(np.abs(r[2] / g["fl_chain_ps"] - 1)) < 1e-6


@pytest.mark.parametrize("name", ["a", "b", "f"])
def test_pertime_oracle_reduces_to_the_reference(name):
    """oracle.gibbs_step_fgmodes_pertime with the SAME flags and noise at every time is the reference's
    step (golden steps.npz): this is what pins the per-time restatement."""
    g = dict(np.load(GOLDEN / "steps.npz"))
    vis, fl, S, F, Ninv, prior = (g[f"{name}_in_{k}"] for k in ("vis", "flags", "S", "fgmodes", "Ninv", "prior"))
    T, N = vis.shape
    flt = np.broadcast_to(fl, (T, N)).copy()
    nt = np.broadcast_to(np.diag(Ninv).real, (T, N)).copy()
    np.random.seed(4242)
    a = R.gibbs_step_fgmodes(vis * fl, fl, S, F, Ninv, prior, solver="direct")
    np.random.seed(4242)
    b = R.gibbs_step_fgmodes_pertime(vis * flt, flt, S, F, nt, prior, solver="direct")
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[0], b[0]) and np.array_equal(a[3], b[3])
    assert np.allclose(a[4], b[4], rtol=1e-14) and b[5] == pytest.approx(a[5], rel=1e-12)
    np.random.seed(4242)
    c = R.gibbs_step_fgmodes_pertime(vis * flt, flt, S, F, nt, prior, solver="cg")
    assert np.max(np.abs(c[2] / g[f"{name}_ps"] - 1)) < 1e-7        # and, with the reference's CG, the golden itself


@pytest.mark.parametrize("flagged", [False, True])
def test_pertime_oracle_with_full_matrices_reduces_to_the_reference(flagged):
    """The (Ntimes, Nfreqs, Nfreqs) form of the per-time oracle with the SAME correlated Ninv (and flags) at every
    time is the reference's dense-noise step: pinned to the golden steps_dense.npz, which the reference itself
    produced (without flags: step_*; with 5 flagged channels: fl_step_*)."""
    g = dict(np.load(GOLDEN / "steps_dense.npz"))
    vis, S, F, Ninv, prior = (g[f"in_{k}"] for k in ("vis", "S", "fgmodes", "Ninv", "prior"))
    T, N = vis.shape
    fl = g["fl_flags"] if flagged else np.ones(N, dtype=bool)
    flt = np.broadcast_to(fl, (T, N)).copy()
    Ninv_t = np.broadcast_to(Ninv, (T, N, N)).copy()
    np.random.seed(4242)
    a = R.gibbs_step_fgmodes(vis * fl, fl, S, F, Ninv, prior, solver="direct")
    np.random.seed(4242)
    b = R.gibbs_step_fgmodes_pertime(vis * flt, flt, S, F, Ninv_t, prior, solver="direct")
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[0], b[0]) and np.array_equal(a[3], b[3])
    assert np.allclose(a[4], b[4], rtol=1e-14) and b[5] == pytest.approx(a[5], rel=1e-12)
    np.random.seed(4242)
    c = R.gibbs_step_fgmodes_pertime(vis * flt, flt, S, F, Ninv_t, prior, solver="cg")
    key = "fl_step_ps" if flagged else "step_ps"
    assert np.max(np.abs(c[2] / g[key] - 1)) < 1e-6
