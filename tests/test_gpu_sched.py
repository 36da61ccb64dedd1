"""The dataflow tile scheduler of the blocked Cholesky sweep (csrc/blocked_sched.inc: chol_sched_kernel, included by blocked.hip; csrc/sched_logic.h) against
the launch-per-phase sweep of rounds 1 - 4: SAME tile code and summation order, so every result must agree bit for bit --
likelihood, beta, status (also of evaluations that fail), prediction tables (extra tile rows), explicit inverse and gradient
(identity rows, lower-triangular extra block), kept factors.  Then the failure path: a schedule that cannot finish must fail
the chunk after its timeout, not hang the device.  Reference anchor: the batch is the grid loop of choose.hyperpars,
Heat Exchanger Emulator/Combined GP Heat Exchanger.R:584-595."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# (CCGP_OPT_SCHED, CCGP_OPT_SCHED_POLICY): 11 = backlog rule + XCD-local synchronisation + chaining (default); 3 = no chaining;
# 9 = agent-scope fences with stealing, chaining; 0 = agent scope, no backlog rule, no chaining; 10 = XCD-local + chaining, both
# workgroups of a CU always take work
VARIANTS = [(1, 11), (2, 11), (1, 3), (2, 3), (1, 9), (2, 0), (1, 10)]


def synth(n, d, seed):
    rng = np.random.default_rng(seed)
    X = rng.uniform(size=(n, d))
    y = np.sin(2 * np.pi * X).sum(axis=1) + 0.1 * rng.normal(size=n)
    return X, y, rng


def draws(rng, B, K, d, rough=20.0):
    P = np.empty((B, K + K * d))
    for b in range(B):
        w = 0.15 + 0.55 * rng.dirichlet(np.ones(K))
        th = np.exp(rng.uniform(np.log(0.5), np.log(50.0), size=(K, d)))
        th[K - 1] = np.maximum(th[K - 1], rough)
        P[b] = np.concatenate([w, th.ravel()])
    return P


def with_sched(handle, sched, policy, fn):
    from ccgp_amd import api
    handle.set_option(api.OPT_SCHED, sched)
    handle.set_option(api.OPT_SCHED_POLICY, policy)
    try:
        return fn()
    finally:
        handle.set_option(api.OPT_SCHED, 3)
        handle.set_option(api.OPT_SCHED_POLICY, 11)


def same(a, b):
    return all(np.array_equal(np.asarray(x), np.asarray(y), equal_nan=True) for x, y in zip(a, b))


@pytest.mark.parametrize("n,d,K,B", [(129, 2, 2, 5), (300, 3, 2, 21), (640, 4, 3, 70), (1000, 5, 3, 37), (2100, 4, 2, 9)])
def test_likelihood_same_bits_as_the_launches(handle, n, d, K, B):
    X, y, rng = synth(n, d, n)
    P = draws(rng, B, K, d)
    for mode, tau2 in ((0, 0.0), (1, 4.0)):
        ref = with_sched(handle, 0, 11, lambda: handle.loglik_batch(X, y, K, P, 1.3, mode, tau2))
        assert np.isfinite(ref[0]).all()
        for sched, policy in VARIANTS:
            got = with_sched(handle, sched, policy, lambda: handle.loglik_batch(X, y, K, P, 1.3, mode, tau2))
            assert same(ref, got), (n, mode, sched, policy)


def test_failing_evaluations_keep_their_status_and_the_others_their_bits(handle):
    """Draws whose matrix is not positive definite (smooth components only, no rough one): NaN and the same pivot index as under
    the launches; the healthy evaluations of the same chunk are untouched."""
    n, d, K, B = 400, 3, 2, 12
    X, y, rng = synth(n, d, 7)
    P = draws(rng, B, K, d, rough=80.0)
    P[3, K:] = 1e-4      # nearly constant correlation: singular to working precision
    P[8, K:] = 1e-4
    ref = with_sched(handle, 0, 11, lambda: handle.loglik_batch(X, y, K, P, 1.0))
    assert (ref[2] != 0).sum() >= 2 and np.isnan(ref[0][3]) and np.isfinite(ref[0][0])
    for sched, policy in VARIANTS:
        got = with_sched(handle, sched, policy, lambda: handle.loglik_batch(X, y, K, P, 1.0))
        assert same(ref, got), (sched, policy)


@pytest.mark.parametrize("n,m", [(257, 40), (700, 129), (1500, 300)])
def test_prediction_tables_same_bits(handle, n, m):
    """Extra tile rows (the cross-correlation rows of predict.post, HX:655-673) ride through the queues like matrix rows."""
    X, y, rng = synth(n, 3, 100 + n)
    P = draws(rng, 6, 2, 3)
    Xt = rng.uniform(size=(m, 3))
    ref = with_sched(handle, 0, 11, lambda: handle.predict_batch(X, y, 2, P, Xt, 2.0))
    for sched, policy in VARIANTS:
        got = with_sched(handle, sched, policy, lambda: handle.predict_batch(X, y, 2, P, Xt, 2.0))
        assert same(ref, got), (n, sched, policy)


@pytest.mark.parametrize("n", [200, 513, 900])
def test_gradient_and_inverse_same_bits(handle, n):
    """Identity rows (solve(R), HX:454, and the gradient): the extra block is lower triangular, rows join at their own block
    column -- the `lower` shapes of sched_logic.h."""
    from ccgp_amd import api
    X, y, rng = synth(n, 2, 200 + n)
    P = draws(rng, 4, 2, 2)
    ref_g = with_sched(handle, 0, 11, lambda: handle.loglik_grad_batch(X, y, 2, P, 1.0))
    th = [np.log(0.7), np.log(25.0), 0.3]
    ref_l = with_sched(handle, 0, 11, lambda: handle.logpost(X, y, 1.0, api.PRIOR_GV, th, want_Rinv=True))
    for sched, policy in VARIANTS:
        got_g = with_sched(handle, sched, policy, lambda: handle.loglik_grad_batch(X, y, 2, P, 1.0))
        assert same(ref_g, got_g), (n, sched, policy)
        got_l = with_sched(handle, sched, policy, lambda: handle.logpost(X, y, 1.0, api.PRIOR_GV, th, want_Rinv=True))
        assert same([ref_l[k] for k in sorted(ref_l)], [got_l[k] for k in sorted(ref_l)]), (n, sched, policy)


def test_kept_factors_from_a_scheduled_sweep_serve_predictions(handle):
    """ccgp_factor_batch under the scheduler, ccgp_predict_from_factorset (rows-only launches) on top of it: the factor a
    scheduled sweep leaves in HBM is the factor the launches leave."""
    n, m = 520, 70
    X, y, rng = synth(n, 3, 31)
    P = draws(rng, 5, 2, 3)
    Xt = rng.uniform(size=(m, 3))
    ref = with_sched(handle, 0, 11, lambda: handle.predict_batch(X, y, 2, P, Xt, 1.5))

    def kept():
        with handle.factor_batch(X, y, 2, P, 1.5) as fs:
            return fs.predict(Xt)
    for sched, policy in ((1, 11), (2, 11), (2, 0)):
        mean, var = with_sched(handle, sched, policy, kept)
        assert np.array_equal(mean, ref[0]) and np.array_equal(var, ref[1]), (sched, policy)


def test_repeated_sweeps_are_deterministic(handle):
    """The order in which workgroups pick tiles differs from run to run; the bits must not.  40 sweeps of 64 matrices (the
    default chunk-size rule picks the scheduler here: n >= 2048, 32 ... 128 matrices) and 40 forced ones at a small size."""
    X, y, rng = synth(2048, 4, 5)
    P = draws(rng, 64, 3, 4)
    ref = with_sched(handle, 0, 11, lambda: handle.loglik_batch(X, y, 3, P, 1.0))
    for _ in range(40):
        assert same(ref, handle.loglik_batch(X, y, 3, P, 1.0))
    X2, y2, rng2 = synth(384, 2, 6)
    P2 = draws(rng2, 150, 2, 2)
    ref2 = with_sched(handle, 0, 11, lambda: handle.loglik_batch(X2, y2, 2, P2, 1.0))
    for rep in range(40):
        assert same(ref2, with_sched(handle, 1 + rep % 2, 11, lambda: handle.loglik_batch(X2, y2, 2, P2, 1.0)))


def test_time_account_adds_up(handle):
    """CCGP_OPT_SCHED_POLICY bit 2: every task of the sweep is accounted to one workgroup, every XCD served its own queue."""
    from ccgp_amd.api import OPT_SCHED, OPT_SCHED_POLICY
    n, B = 1024, 48
    X, y, rng = synth(n, 3, 9)
    P = draws(rng, B, 2, 3)
    nt = n // 128
    tasks = nt + sum(nt - 1 - j for j in range(1, nt)) + sum(nt - j for j in range(nt))
    for sched in (1, 2):
        handle.set_option(OPT_SCHED, sched)
        handle.set_option(OPT_SCHED_POLICY, 11 | 4)
        try:
            handle.loglik_batch(X, y, 2, P, 1.0)
            acc = handle.last_sched_profile()
        finally:
            handle.set_option(OPT_SCHED, 3)
            handle.set_option(OPT_SCHED_POLICY, 11)
        assert acc.shape[0] in (256 * (3 - sched), 304 * (3 - sched)) or acc.shape[0] > 0
        assert acc[:, 5].sum() == tasks * B
        for q in range(8):      # XCD-local synchronisation: a queue is served by its own XCD's workgroups only
            assert acc[acc[:, 6] == q, 5].sum() == tasks * len(range(q, B, 8))


def test_a_schedule_that_cannot_finish_fails_the_chunk_and_returns(handle):
    """Test hook (policy bit 4): the announcements of matrix 0's second block column are dropped, so the sweep can never
    complete.  Every workgroup must leave after the timeout, every evaluation of the chunk must come back failed (NaN, status
    != 0) -- and the next call on the same handle must work."""
    from ccgp_amd import api
    old = os.environ.get("CCGP_SCHED_TIMEOUT_MS")
    os.environ["CCGP_SCHED_TIMEOUT_MS"] = "300"
    try:
        h = api.Handle(0)
    finally:
        if old is None:
            del os.environ["CCGP_SCHED_TIMEOUT_MS"]
        else:
            os.environ["CCGP_SCHED_TIMEOUT_MS"] = old
    try:
        X, y, rng = synth(640, 3, 77)
        P = draws(rng, 11, 2, 3)
        h.set_option(api.OPT_SCHED, 1)
        h.set_option(api.OPT_SCHED_POLICY, 11 | 16)
        ll, beta, st = h.loglik_batch(X, y, 2, P, 1.0)
        assert np.isnan(ll).all() and (st != 0).all()
        h.set_option(api.OPT_SCHED_POLICY, 11)
        ll2, _, st2 = h.loglik_batch(X, y, 2, P, 1.0)
        ref = with_sched(handle, 0, 11, lambda: handle.loglik_batch(X, y, 2, P, 1.0))
        assert not st2.any() and np.array_equal(ll2, ref[0])
    finally:
        h.close()
