"""The split factorisation (several co-operating workgroups per system, hydra_pspec_amd/csrc/hpx_factor_split.hip)
under the conditions its hand-off protocol has to survive: two plans on two streams at once, a forced time-out, the
agent-scope fall-back protocol, and the switch that turns the form off."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

T_, M_ = 32, 12


def _batch(nbl, N, k0, niter, **kw):
    from hydra_pspec_amd import pspec, synthetic
    d = synthetic.make_baselines(N, T_, M_, k0=k0, nbl=nbl, flag_frac=0.0, dense=False)
    gb = pspec.GibbsBatch(d["vis"], d["flags"], d["fgmodes"], d["ninv_diag"], d["ps_prior"], niter, seed=d["seed"],
                          solver="dense", **kw)
    ps0 = np.ascontiguousarray(np.broadcast_to(np.asarray(d["ps0"], dtype=float), (nbl, N)))
    return gb, ps0


def _chain(gb, ps0, niter):
    gb.iter_done = 0
    out = gb.run(niter, ps0=ps0)
    return out["signal_ps"].cpu().numpy(), out["ln_post"].cpu().numpy()


def test_two_small_plans_on_two_streams_at_once():
    """Two 8-baseline plans of order 272 (eight workgroups per system each: 2 x 64 workgroups, all resident together)
    run 50 iterations CONCURRENTLY on two streams from two host threads: each gets the chain it gets alone, bit for
    bit (the launcher keeps count of the split launches in flight per device)."""
    import torch
    from hydra_pspec_amd import hpx
    niter, N = 50, 256
    assert hpx.lib().hpx_version() > 0
    plans = [_batch(8, N, k0, niter) for k0 in (0, 40)]
    alone = [_chain(gb, ps0, niter) for gb, ps0 in plans]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got, errs = [None, None], []
    start = threading.Barrier(2)

    def work(i):
        try:
            with torch.cuda.stream(streams[i]):
                start.wait()
                got[i] = _chain(plans[i][0], plans[i][1], niter)
        except BaseException as e:      # noqa: BLE001
            errs.append(e)

    for rep in range(2):
        th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs, errs
        for i in range(2):
            assert np.array_equal(got[i][0], alone[i][0]) and np.array_equal(got[i][1], alone[i][1]), (rep, i)
    torch.cuda.synchronize()
    for gb, _ in plans:
        gb.close()


def test_more_concurrent_plans_than_the_device_holds():
    """Three 16-baseline plans of order 272 want 3 x 128 workgroups of the split form: one more than the device has
    CUs for.  The launcher gives the late-comer fewer workgroups per system or the one-workgroup kernel instead of
    letting its parts wait for CUs the others hold: every chain completes and agrees with its single-stream chain to
    rounding (bit for bit is not promised across forms)."""
    import torch
    niter, N = 12, 256
    plans = [_batch(16, N, k0, niter) for k0 in (0, 20, 50)]
    alone = [_chain(gb, ps0, niter) for gb, ps0 in plans]
    streams = [torch.cuda.Stream() for _ in plans]
    got, errs = [None] * 3, []
    start = threading.Barrier(3)

    def work(i):
        try:
            with torch.cuda.stream(streams[i]):
                start.wait()
                got[i] = _chain(plans[i][0], plans[i][1], niter)
        except BaseException as e:      # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for i in range(3):
        assert np.max(np.abs(got[i][0] / alone[i][0] - 1)) < 1e-6, i
    torch.cuda.synchronize()
    for gb, _ in plans:
        gb.close()


def test_a_handoff_timeout_has_its_own_error():
    """A spin that gives up is reported as HPX_ETIMEOUT / hpx.HpxTimeout with bit 30 of the baseline's info word --
    not as a non-positive pivot (FloatingPointError).  Forced by letting every spin give up after one poll, with the
    automatic repeat (HPX_OPT_SPLIT_RETRY) switched off."""
    from hydra_pspec_amd import hpx
    gb, ps0 = _batch(8, 256, 3, 4)
    hpx.set_option(hpx.OPT_SPLIT_RETRY, 0, plan=gb.plan)
    hpx.set_option(hpx.OPT_SPLIT_SPIN_LIMIT, 1)
    try:
        with pytest.raises(hpx.HpxTimeout, match="timed out"):
            _chain(gb, ps0, 4)
        info = gb.plan.info()
        assert (info & hpx.INFO_TIMEOUT).any()
    finally:
        hpx.set_option(hpx.OPT_SPLIT_SPIN_LIMIT, 0)        # (0 = the default limit)
        gb.close()
    # the library is fine afterwards: the counters of the timed-out systems were left zero
    gb, ps0 = _batch(8, 256, 3, 4)
    a = _chain(gb, ps0, 4)
    b = _chain(gb, ps0, 4)
    gb.close()
    assert np.isfinite(a[0]).all() and np.array_equal(a[0], b[0])


def test_a_timed_out_run_is_repeated_without_the_split_form():
    """Default (HPX_OPT_SPLIT_RETRY = 1): hpx_gibbs_run repeats a run whose split factor timed out on the one-workgroup
    kernel and the plan leaves the form for good -- the caller gets the chain of allow_split=False, bit for bit, for a
    run started from given bandpowers and for one that continues from the plan's own; the fall-back is counted."""
    from hydra_pspec_amd import hpx
    niter = 3
    gb, ps0 = _batch(8, 256, 3, 2 * niter, allow_split=False)
    want1 = _chain(gb, ps0, niter)
    out = gb.run(niter)                                    # continues from the plan's bandpowers
    want2 = out["signal_ps"].cpu().numpy()
    gb.close()
    # (a) the first run times out
    gb, ps0 = _batch(8, 256, 3, 2 * niter)
    assert hpx.get_option(hpx.OPT_FACTOR_SPLIT, gb.plan) == 1 and hpx.get_option(hpx.OPT_SPLIT_FALLBACKS, gb.plan) == 0
    hpx.set_option(hpx.OPT_SPLIT_SPIN_LIMIT, 1)
    try:
        got1 = _chain(gb, ps0, niter)
    finally:
        hpx.set_option(hpx.OPT_SPLIT_SPIN_LIMIT, 0)
    assert hpx.get_option(hpx.OPT_FACTOR_SPLIT, gb.plan) == 0 and hpx.get_option(hpx.OPT_SPLIT_FALLBACKS, gb.plan) == 1
    assert np.array_equal(got1[0], want1[0]) and np.array_equal(got1[1], want1[1])
    got2 = gb.run(niter)["signal_ps"].cpu().numpy()
    assert np.array_equal(got2, want2)
    gb.close()
    # (b) the continuing run times out: it is repeated from the bandpowers it started from -- the same as switching the
    # form off by hand between the two runs
    gb, ps0 = _batch(8, 256, 3, 2 * niter)
    first = _chain(gb, ps0, niter)
    hpx.set_option(hpx.OPT_FACTOR_SPLIT, 0, plan=gb.plan)
    want = gb.run(niter)["signal_ps"].cpu().numpy()
    gb.close()
    gb, ps0 = _batch(8, 256, 3, 2 * niter)
    again = _chain(gb, ps0, niter)
    assert np.array_equal(first[0], again[0])
    hpx.set_option(hpx.OPT_SPLIT_SPIN_LIMIT, 1)
    try:
        cont = gb.run(niter)["signal_ps"].cpu().numpy()
    finally:
        hpx.set_option(hpx.OPT_SPLIT_SPIN_LIMIT, 0)
    assert hpx.get_option(hpx.OPT_SPLIT_FALLBACKS, gb.plan) == 1
    assert np.array_equal(cont, want)
    gb.close()


def test_agent_scope_handoff_protocol():
    """The protocol the split form falls back to when the parts of a system do not share an XCD (agent-scope release /
    acquire fences), forced through HPX_OPT_SPLIT_HEAVY: zpotrs of a small batch against numpy, and the chain of a
    small plan bit for bit the one of the light protocol."""
    import torch
    from hydra_pspec_amd import hpx
    rng = np.random.default_rng(3)
    nb, n, nrhs = 5, 200, 24
    a = rng.standard_normal((nb, n, n)) + 1j * rng.standard_normal((nb, n, n))
    A = a @ np.conj(np.swapaxes(a, 1, 2)) / n + np.eye(n)
    B = rng.standard_normal((nb, n, nrhs)) + 1j * rng.standard_normal((nb, n, nrhs))
    gb, ps0 = _batch(8, 256, 9, 3)
    light = _chain(gb, ps0, 3)
    hpx.set_option(hpx.OPT_SPLIT_HEAVY, 1)
    try:
        dA, dB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
        dX = torch.zeros_like(dB)
        info = torch.zeros(nb, dtype=torch.int32, device="cuda")
        hpx.check(hpx.lib().hpx_zpotrs_batched(nb, n, nrhs, hpx.ptr(dA), hpx.ptr(dB), hpx.ptr(dX), hpx.ptr(info), None))
        X = dX.cpu().numpy()
        assert not info.cpu().numpy().any()
        assert np.abs(X - np.linalg.solve(A, B)).max() / np.abs(X).max() < 1e-11
        heavy = _chain(gb, ps0, 3)
    finally:
        hpx.set_option(hpx.OPT_SPLIT_HEAVY, 0)
        gb.close()
    assert np.array_equal(light[0], heavy[0]) and np.array_equal(light[1], heavy[1])


def test_split_switched_off_gives_the_large_batch_chain():
    """allow_split=False (HPX_OPT_FACTOR_SPLIT = 0 on the plan): a small batch then takes the one-workgroup kernel and a
    baseline's chain is bit for bit the one it has inside a batch large enough never to split -- what a driver that
    re-shards between runs relies on; with the split form on they agree to rounding."""
    niter, N = 3, 256
    gb, ps0 = _batch(8, N, 0, niter, allow_split=False)
    small_off = _chain(gb, ps0, niter)
    gb.close()
    gb, ps0 = _batch(8, N, 0, niter)
    small_on = _chain(gb, ps0, niter)
    gb.close()
    gb, ps0 = _batch(136, N, 0, niter)
    big = _chain(gb, ps0, niter)
    gb.close()
    assert np.array_equal(small_off[0], big[0][:8]) and np.array_equal(small_off[1], big[1][:8])
    assert np.max(np.abs(small_on[0] / big[0][:8] - 1)) < 1e-6
    from hydra_pspec_amd import hpx
    assert hpx.lib().hpx_set_option(None, 12345, 1) == hpx.HPX_EINVAL
