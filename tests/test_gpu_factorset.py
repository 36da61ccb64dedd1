"""Device-resident factor set (ccgp_factor_batch / ccgp_predict_from_factorset): predictions from
kept factors are the predictions of ccgp_predict_batch, for any number of test sets, without re-factorising."""
import numpy as np
import pytest

from conftest import load_gv, synthetic_design
from oracle import ccgp_oracle as orc

pytestmark = pytest.mark.gpu


def test_n90_fused_path_two_test_sets(handle):
    D, y, Dt, _ = load_gv(90)
    draws = np.array([[0.6 + 0.05 * s, 0.25 + 0.02 * s, 12.0 + 2 * s] for s in range(5)])
    P = np.array([orc.params_from_iso(p, t1, t2, 9) for p, t1, t2 in draws])
    ll, beta, st = handle.loglik_batch(D, y, 2, P, 10.0)
    with handle.factor_batch(D, y, 2, P, 10.0) as fs:
        np.testing.assert_array_equal(fs.loglik, ll)
        np.testing.assert_array_equal(fs.beta, beta)
        assert not fs.status.any()
        for sites in (Dt[:40], Dt[40:110]):
            want = handle.predict_batch(D, y, 2, P, sites, 10.0)
            got = fs.predict(sites)
            np.testing.assert_array_equal(got[0], want[0])
            np.testing.assert_array_equal(got[1], want[1])
    # and against the oracle (reference arithmetic) for one site set
    wm, wv, _ = orc.predict_table(D, y, draws[:2], Dt[:5], 10.0)
    with handle.factor_batch(D, y, 2, P[:2], 10.0) as fs:
        m, v = fs.predict(Dt[:5])
    np.testing.assert_allclose(m, wm, rtol=1e-8)
    np.testing.assert_allclose(v, wv, rtol=1e-6, atol=1e-9)


def test_n520_blocked_path_two_test_sets(handle):
    X, y = synthetic_design(520, 3, 11)
    rng = np.random.default_rng(2)
    P = np.array([np.concatenate([[0.6, 0.4], np.exp(rng.uniform(0.0, 1.5, 3)), np.exp(rng.uniform(3.0, 4.0, 3))])
                  for _ in range(6)])
    sets = (rng.random((37, 3)), rng.random((200, 3)))               # one and two extra tile rows
    ll, beta, st = handle.loglik_batch(X, y, 2, P, 1.0)
    with handle.factor_batch(X, y, 2, P, 1.0) as fs:
        assert fs.nbytes > 6 * 640 * 640 * 8
        np.testing.assert_array_equal(fs.loglik, ll)
        np.testing.assert_array_equal(fs.beta, beta)
        for sites in sets + (sets[0],):                              # a set may be served again
            want = handle.predict_batch(X, y, 2, P, sites, 1.0)
            got = fs.predict(sites)
            # same tile kernels, same k order as the full sweep: bit-identical
            np.testing.assert_array_equal(got[0], want[0])
            np.testing.assert_array_equal(got[1], want[1])
        # interleaving other work on the handle does not disturb the kept factors
        handle.loglik_batch(X, y, 2, P[:2], 2.0)
        np.testing.assert_array_equal(fs.predict(sets[0])[0], handle.predict_batch(X, y, 2, P, sets[0], 1.0)[0])
        # a workspace limit that holds the scratch rows of two draws only (200 sites -> 256 x 640 doubles per draw):
        # the sweep over the kept factors runs in chunks of draws (round-2 advisor) -- same bits, any chunking
        whole = fs.predict(sets[1])
        handle.set_workspace_limit(2 * 256 * 640 * 8 + 4096)
        try:
            chunked = fs.predict(sets[1])
        finally:
            handle.set_workspace_limit(200 << 30)
        np.testing.assert_array_equal(chunked[0], whole[0])
        np.testing.assert_array_equal(chunked[1], whole[1])
    # oracle for one draw and a few sites
    w, Th = orc.unpack_params(P[0], 2, 3)
    R = orc.mixed_corr_matrix_general(X, w, Th)
    R_inv = orc.solve_inverse(R)
    b = orc.beta_mle(R_inv, y)
    mf, v1, v2 = orc.factors(R_inv, b, y)
    with handle.factor_batch(X, y, 2, P[:1], 1.0) as fs:
        m, v = fs.predict(sets[0][:4])
    for t in range(4):
        wm, wv = orc.predict_post_from_factors(orc.mixed_corr_vec_general(sets[0][t], X, w, Th), b, mf, v1, v2, R_inv, 1.0)
        assert m[0, t] == pytest.approx(wm, rel=1e-8) and v[0, t] == pytest.approx(wv, rel=1e-5, abs=1e-9)


def test_failed_draw_is_nan_in_every_table(handle):
    X, y = synthetic_design(300, 2, 5)
    X[7] = X[3]                                                      # duplicated site: singular R
    P = np.array([[0.5, 0.5, 1.0, 1.0, 2.0, 2.0]])
    with handle.factor_batch(X, y, 2, P, 1.0) as fs:
        assert fs.status[0] != 0 and np.isnan(fs.loglik[0])
        m, v = fs.predict(np.array([[0.3, 0.3], [0.6, 0.1]]))
        assert np.isnan(m).all() and np.isnan(v).all()


def test_matern_family_factor_set(handle):
    """The 1-D script's Matern family always takes the materialised-matrix path: a factor set works there too
    (n = 8 -> one 128 x 128 tile), with the family captured at factorisation time."""
    from ccgp_amd import api
    X = np.linspace(0.05, 0.95, 8)[:, None]
    y = np.sin(10.0 * X[:, 0])
    P = np.array([[0.6, 0.4, 0.3, 0.08], [0.75, 0.25, 0.4, 0.12], [0.5, 0.5, 0.25, 0.1]])
    sites = np.linspace(0.0, 1.0, 140)[:, None]                      # two extra tile rows
    try:
        handle.set_kernel(api.KERNEL_MATERN, 5.0)
        want = handle.predict_batch(X, y, 2, P, sites, 0.7)
        fs = handle.factor_batch(X, y, 2, P, 0.7)
    finally:
        handle.set_kernel(api.KERNEL_GAUSS, 0.0)
    with fs:
        got = fs.predict(sites)                                       # the handle is back on the Gaussian family
        np.testing.assert_array_equal(got[0], want[0])
        np.testing.assert_array_equal(got[1], want[1])
