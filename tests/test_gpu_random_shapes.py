"""Seeded random shapes through every batched entry point of the C ABI, each checked against the oracle.

The parity tests pin the reference's own designs and the sizes the kernels were tuned for; this sweep walks the
dispatch boundaries instead (n around 8 k, 64, 104, 128, 129 and tile multiples; d = 1 ... 12; K = 1 ... 4; one test site
and many; batches that are not multiples of anything) with well-conditioned draws, so that any instantiation, LDS
carve-up or ragged edge that the tuned sizes never reach is executed at least once.  Tolerances follow the condition
number of the draw's matrix: fp64 delivers cond * eps, in R as here.
"""
import numpy as np
import pytest

from conftest import synthetic_design
from oracle import ccgp_oracle as orc

pytestmark = pytest.mark.gpu

SIZES = [5, 8, 9, 31, 63, 64, 65, 72, 97, 104, 105, 127, 128, 129, 130, 200, 255, 256, 257, 300, 384, 385]


def draws(rng, n, d, K, B):
    """Components from smooth to rough in the units of the design's spacing; the roughest keeps R well conditioned."""
    rough = 2.0 * n ** (2.0 / d) / d
    P = np.empty((B, K + K * d))
    for b in range(B):
        th = np.exp(rng.uniform(np.log(0.02 * rough), np.log(0.3 * rough), size=(K, d)))
        th[-1] = rng.uniform(rough, 2.0 * rough, d)
        P[b] = np.concatenate([0.2 + 0.6 * rng.dirichlet(np.ones(K)), th.ravel()])
    return P


def cases(seed, count):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        n = int(SIZES[(i * 7 + int(rng.integers(0, 3))) % len(SIZES)])
        d = int(rng.choice([1, 2, 3, 4, 5, 6, 8, 9, 12]))
        K = int(rng.choice([1, 2, 2, 3, 3, 4]))
        out.append((n, d, K, int(rng.integers(0, 2 ** 31))))
    return out


@pytest.mark.parametrize("n,d,K,seed", cases(20260401, 28))
def test_likelihood_both_mean_modes(handle, n, d, K, seed):
    from ccgp_amd import api
    rng = np.random.default_rng(seed)
    X, y = synthetic_design(n, d, seed)
    B = int(rng.choice([1, 3, 9, 66]))
    P = draws(rng, n, d, K, B)
    sigma2 = float(rng.uniform(0.3, 3.0))
    for mode, tau2 in ((api.MEAN_PROFILE_BETA, 0.0), (api.MEAN_ZERO_PLUS_TAU2, float(rng.uniform(0.5, 20.0)))):
        ll, beta, st = handle.loglik_batch(X, y, K, P, sigma2, mode, tau2)
        assert not st.any(), (n, d, K, B, mode)
        for b in {0, B // 2, B - 1}:
            w, Th = orc.unpack_params(P[b], K, d)
            cond = np.linalg.cond(orc.mixed_corr_matrix_general(X, w, Th))
            want_ll, want_beta = orc.loglik_general(X, y, w, Th, sigma2, mode, tau2)
            assert ll[b] == pytest.approx(want_ll, rel=max(1e-10, 50 * cond * np.finfo(float).eps)), (n, d, K, B, mode, cond)
            if mode == api.MEAN_PROFILE_BETA:
                assert beta[b] == pytest.approx(want_beta, rel=max(1e-9, 100 * cond * np.finfo(float).eps), abs=1e-9)


@pytest.mark.parametrize("n,d,K,seed", cases(20260402, 18))
def test_prediction_tables(handle, n, d, K, seed):
    rng = np.random.default_rng(seed)
    X, y = synthetic_design(n, d, seed)
    S = int(rng.choice([1, 2, 7]))
    m = int(rng.choice([1, 2, 30, 31, 62, 63, 129]))
    P = draws(rng, n, d, K, S)
    Xt = rng.random((m, d))
    sigma2 = float(rng.uniform(0.3, 3.0))
    mean, var, beta, st = handle.predict_batch(X, y, K, P, Xt, sigma2)
    assert not st.any() and mean.shape == (S, m) and var.shape == (S, m)
    for s in {0, S - 1}:
        w, Th = orc.unpack_params(P[s], K, d)
        R = orc.mixed_corr_matrix_general(X, w, Th)
        cond = np.linalg.cond(R)
        R_inv = orc.solve_inverse(R)
        b_ = orc.beta_mle(R_inv, y)
        mf, v1, v2 = orc.factors(R_inv, b_, y)
        tol = max(1e-8, 200 * cond * np.finfo(float).eps)
        assert beta[s] == pytest.approx(b_, rel=tol, abs=1e-9)
        for t in {0, m // 2, m - 1}:
            r = orc.mixed_corr_vec_general(Xt[t], X, w, Th)
            want = orc.predict_post_from_factors(r, b_, mf, v1, v2, R_inv, sigma2)
            assert mean[s, t] == pytest.approx(want[0], rel=tol, abs=tol), (n, d, K, m, cond)
            assert var[s, t] == pytest.approx(want[1], rel=10 * tol, abs=10 * tol * sigma2), (n, d, K, m, cond)


@pytest.mark.parametrize("n,d,K,seed", cases(20260403, 18))
def test_gradient_against_central_differences_of_the_device_likelihood(handle, n, d, K, seed):
    rng = np.random.default_rng(seed)
    X, y = synthetic_design(n, d, seed)
    B = int(rng.choice([1, 2, 5]))
    rows = draws(rng, n, d, K, B)
    sigma2 = float(rng.uniform(0.3, 3.0))
    ll, beta, grad, st = handle.loglik_grad_batch(X, y, K, rows, sigma2)
    assert not st.any() and np.isfinite(grad).all()
    ll0, beta0, _ = handle.loglik_batch(X, y, K, rows, sigma2)
    np.testing.assert_allclose(ll, ll0, rtol=1e-11)      # gradient instances re-derive the value (bitwise where they share the kernel)
    P = rows.shape[1]
    b = B - 1
    pert = np.repeat(rows[b][None], 2 * P, axis=0)
    hstep = 1e-5 * np.abs(rows[b])
    for j in range(P):
        pert[2 * j, j] += hstep[j]
        pert[2 * j + 1, j] -= hstep[j]
    llp, _, stp = handle.loglik_batch(X, y, K, pert, sigma2)
    assert not stp.any()
    fd = (llp[0::2] - llp[1::2]) / (2 * hstep)
    np.testing.assert_allclose(grad[b], fd, rtol=5e-4, atol=5e-4 * np.abs(fd).max(), err_msg="n=%d d=%d K=%d" % (n, d, K))


@pytest.mark.parametrize("n,d,K,seed", cases(20260404, 10))
def test_kept_factors_and_design_logdets(handle, n, d, K, seed):
    """ccgp_factor_batch + ccgp_predict_from_factorset equal ccgp_predict_batch bit for bit at any shape; the log-determinants
    of candidate designs (entropy criteria, BSQ:856-877) equal numpy's slogdet."""
    rng = np.random.default_rng(seed)
    X, y = synthetic_design(n, d, seed)
    S, m = 3, int(rng.choice([1, 17, 140]))
    P = draws(rng, n, d, K, S)
    Xt = rng.random((m, d))
    a = handle.predict_batch(X, y, K, P, Xt, 1.1)
    with handle.factor_batch(X, y, K, P, 1.1) as fs:
        mean, var = fs.predict(Xt)
    np.testing.assert_array_equal(mean, a[0])
    np.testing.assert_array_equal(var, a[1])
    designs = np.stack([synthetic_design(n, d, seed + 1 + i)[0] for i in range(4)])
    ld, st = handle.mixed_logdet_designs(designs, K, P[0])
    assert not np.any(st)
    w, Th = orc.unpack_params(P[0], K, d)
    for i in range(4):
        want = np.linalg.slogdet(orc.mixed_corr_matrix_general(designs[i], w, Th))[1]
        assert ld[i] == pytest.approx(want, rel=1e-8, abs=1e-7), (n, d, K, i)


@pytest.mark.parametrize("n,nu,seed", [(7, 1.25, 1), (33, 1.5, 2), (64, 2.5, 3), (90, 5.0, 4), (128, 1.5, 5), (129, 2.5, 6),
                                        (200, 7.5, 7), (300, 5.0, 8)])
def test_matern_family_on_both_paths(handle, n, nu, seed):
    """The 1-D scripts' Matern kernel (D1:348-351) through the batched likelihood and prediction, on the register-resident
    evaluator (n <= 128) and on the blocked sweep; design points a jittered grid on [0, 1], length scales around the spacing
    (smoother kernels on closer points are numerically singular at any precision)."""
    from ccgp_amd import api
    rng = np.random.default_rng(seed)
    x = np.sort((np.arange(n) + rng.uniform(0.2, 0.8, n)) / n)[:, None]
    y = np.sin(9.0 * x[:, 0]) + 0.3 * np.cos(31.0 * x[:, 0])
    h = 1.0 / n
    B = 4
    P = np.column_stack([rng.uniform(0.3, 0.9, B), rng.uniform(0.1, 0.7, B), rng.uniform(1.5, 3.0, B) * h,
                         rng.uniform(0.3, 0.8, B) * h])
    Xt = rng.random((5, 1))
    try:
        handle.set_kernel(api.KERNEL_MATERN, nu)
        ll, beta, st = handle.loglik_batch(x, y, 2, P, 1.7)
        mean, var, beta2, st2 = handle.predict_batch(x, y, 2, P[:2], Xt, 1.7)
    finally:
        handle.set_kernel(api.KERNEL_GAUSS, 0.0)
    assert not st.any() and not st2.any()
    for b in range(B):
        w = P[b, :2]
        R = (w[0] ** 2 * orc.corr_matrix_matern(nu, x, P[b, 2]) + w[1] ** 2 * orc.corr_matrix_matern(nu, x, P[b, 3])) / np.sum(w ** 2)
        cond = np.linalg.cond(R)
        R_inv = orc.solve_inverse(R)
        b_ = orc.beta_mle(R_inv, y)
        want = orc.dmnorm_log(y, b_, 1.7 * np.sum(w ** 2) * R)
        tol = max(1e-9, 100 * cond * np.finfo(float).eps)
        assert ll[b] == pytest.approx(want, rel=tol), (n, nu, cond)
        assert beta[b] == pytest.approx(b_, rel=10 * tol, abs=1e-9)
        if b < 2:
            mf, v1, v2 = orc.factors(R_inv, b_, y)
            for t in range(5):
                r = (w[0] ** 2 * orc.corr_vec_matern(Xt[t, 0], x, P[b, 2], nu) +
                     w[1] ** 2 * orc.corr_vec_matern(Xt[t, 0], x, P[b, 3], nu)) / np.sum(w ** 2)
                wm, wv = orc.predict_post_from_factors(r, b_, mf, v1, v2, R_inv, 1.7)
                assert mean[b, t] == pytest.approx(wm, rel=10 * tol, abs=10 * tol)
                assert var[b, t] == pytest.approx(wv, rel=100 * tol, abs=100 * tol * 1.7)


@pytest.mark.parametrize("n,d,G,N,take_log,seed", [(12, 2, 5, 16, True, 1), (64, 4, 3, 33, True, 2), (100, 2, 4, 20, False, 3),
                                                     (129, 3, 2, 9, True, 4), (260, 2, 3, 7, True, 5)])
def test_hyperprior_grid_random_shapes(handle, n, d, G, N, take_log, seed):
    """choose.hyperpars (HX:584-595 / ADV:588-599) for grids, node counts and designs other than the scripts' own: the
    device builds the G x N draws (Halton nodes, inverse-gamma quantiles) and averages; the oracle walks the R loops."""
    rng = np.random.default_rng(seed)
    X, y = synthetic_design(n, d, seed)
    s = n ** (2.0 / d) / d                                   # theta ~ s: correlation e^-1 at the typical spacing
    hyper = np.column_stack([rng.uniform(3.0, 6.0, G), rng.uniform(0.3, 0.8, G) * s * 4.0,
                             rng.uniform(3.0, 6.0, G), rng.uniform(2.0, 5.0, G) * s * 4.0])
    tau = 7.0
    vals, arg = handle.grid_marginal(X, y, 1.3, hyper, N, tau, take_log)
    want_arg, want = orc.choose_hyperpars(X, y, hyper, 1.3, N, tau, take_log)
    np.testing.assert_allclose(vals, want, rtol=1e-8)
    assert arg == want_arg                                    # 0-based row (the R shim adds the 1)
