"""Pins for the CPU oracle (oracle/ccgp_oracle.py).

The reference has no golden vectors and cannot run here (no R), so the oracle is
"parity unpinned" with respect to the reference itself; what CAN be pinned is:
  * agreement with an independent 50-digit mpmath evaluation (different formulation),
  * analytic properties of the model,
  * stability of the committed fixtures (tests/golden/*.json).
"""
import math

import numpy as np
import pytest

from conftest import golden, load_gv, load_hyper, load_maximin, load_qian
from oracle import ccgp_oracle as orc
from oracle import mp_check


def test_gram_matrix_properties():
    D, y, _, _ = load_qian()
    R = orc.corr_matrix(D, [0.3, 1.1, 2.0, 0.7])
    np.testing.assert_allclose(R, R.T, rtol=0, atol=1e-15)
    np.testing.assert_allclose(np.diag(R), 1.0, rtol=0, atol=1e-14)   # expanded form: not exactly 1
    assert np.all(R > 0) and np.all(R <= 1 + 1e-14)
    # direct squared-difference form agrees (HX:352-355 is only an expansion of it)
    diff = D[:, None, :] - D[None, :, :]
    direct = np.exp(-(diff ** 2 * np.array([0.3, 1.1, 2.0, 0.7])).sum(-1))
    np.testing.assert_allclose(R, direct, rtol=1e-13)
    np.testing.assert_allclose(orc.corr_matrix_iso(D, 0.45), orc.corr_matrix(D, [0.45] * 4), rtol=0, atol=0)


def test_corr_vec_is_a_gram_row():
    D, _, Dt, _ = load_qian()
    full = orc.corr_matrix_iso(np.vstack([Dt[:1], D]), 0.8)
    np.testing.assert_allclose(orc.corr_vec_iso(Dt[0], D, 0.8), full[0, 1:], rtol=1e-13)


def test_mix_limits():
    D, _, _, _ = load_qian()
    np.testing.assert_allclose(orc.mixed_corr_matrix_iso(D, 1.0, 0.3, 9.0), orc.corr_matrix_iso(D, 0.3), rtol=1e-15)
    np.testing.assert_allclose(orc.mixed_corr_matrix_iso(D, 0.0, 0.3, 9.0), orc.corr_matrix_iso(D, 9.0), rtol=1e-15)
    D2 = load_maximin(14)
    np.testing.assert_allclose(orc.mixed_corr_matrix_aniso(D2, 0.7, 0.5, 0.9, 0.0),
                               orc.corr_matrix(D2, [0.5, 0.9]), rtol=1e-14)
    w, Th = orc.unpack_params(orc.params_from_iso(0.7, 0.3, 15.0, 4), 2, 4)
    np.testing.assert_allclose(orc.mixed_corr_matrix_general(D, w, Th),
                               orc.mixed_corr_matrix_iso(D, 0.7, 0.3, 15.0), rtol=1e-14)


def test_logpost_decomposition_and_invariances():
    D, y, _, _ = load_qian()
    s2 = 10.0
    t = [math.log(0.3), math.log(15.0), math.log(0.8 / 0.2)]
    lp = orc.logpost(D, t, y, s2, "HX", (7, 3, 3, 28))
    assert lp["val"] == pytest.approx(lp["log_like"] + orc.log_jacobian(t) + orc.log_prior(t, "HX", (7, 3, 3, 28)), rel=1e-15)
    # profiled intercept: shifting y shifts beta and leaves the likelihood alone
    lp2 = orc.logpost(D, t, y + 3.5, s2, "HX", (7, 3, 3, 28))
    assert lp2["beta"] == pytest.approx(lp["beta"] + 3.5, rel=1e-11)
    assert lp2["log_like"] == pytest.approx(lp["log_like"], rel=1e-10)
    # R.Inv really is the inverse; beta.MLE / sigma2.MLE agree with their definitions
    R = orc.mixed_corr_matrix_iso(D, 0.8, 0.3, 15.0)
    np.testing.assert_allclose(lp["R_inv"] @ R, np.eye(64), atol=1e-9)
    one = np.ones(64)
    assert orc.beta_mle(lp["R_inv"], y) == pytest.approx((one @ np.linalg.solve(R, y)) / (one @ np.linalg.solve(R, one)), rel=1e-10)
    u = y - lp["beta"]
    assert orc.sigma2_mle(lp["R_inv"], y, lp["beta"]) == pytest.approx(u @ np.linalg.solve(R, u) / 64, rel=1e-9)
    # general-K form reproduces the likelihood term
    w, Th = orc.unpack_params(orc.params_from_iso(0.8, 0.3, 15.0, 4), 2, 4)
    ll, beta = orc.loglik_general(D, y, w, Th, s2)
    assert ll == pytest.approx(lp["log_like"], rel=1e-13) and beta == pytest.approx(lp["beta"], rel=1e-13)


def test_priors_match_each_script():
    t3 = [0.2, 1.5, -0.4]
    th1, th2 = math.exp(0.2), math.exp(1.5)
    assert orc.log_prior(t3, "GV") == pytest.approx(-4 * 0.2 - 1 / th1 - 6 * 1.5 - 75 / th2)
    assert orc.log_prior(t3, "ISO") == pytest.approx(-4 * 0.2 - 2 / th1 - 6 * 1.5 - 16 / th2)
    assert orc.log_prior(t3, "BSQ") == orc.log_prior(t3, "ISO") == orc.log_prior(t3, "D1")
    assert orc.log_prior(t3, "HX", (7, 3, 3, 28)) == pytest.approx(-8 * 0.2 - 3 / th1 - 4 * 1.5 - 28 / th2)
    t4 = t3 + [0.7]
    assert orc.log_prior(t4, "ANI") == pytest.approx(-0.2 - 0.02 - 1.5 - 1.125 - 2.8 - 4 / math.exp(0.7))
    assert orc.log_jacobian(t4) == pytest.approx(orc.log_jacobian(t3) + 0.7)


def test_dmnorm_against_scipy():
    import scipy.stats as sst
    rng = np.random.default_rng(3)
    A = rng.normal(size=(9, 9))
    S = A @ A.T + 9 * np.eye(9)
    x = rng.normal(size=9)
    assert orc.dmnorm_log(x, 0.3, S) == pytest.approx(sst.multivariate_normal(np.full(9, 0.3), S).logpdf(x), rel=1e-12)
    with pytest.raises(np.linalg.LinAlgError):
        orc.dmnorm_log(x, 0.0, -S)


@pytest.mark.parametrize("mode", [0, 1])
def test_oracle_vs_mpmath_maximin14(mode):
    D = load_maximin(14)
    y = np.array([orc.test_function_2d(a, b, 3) for a, b in D])
    for row, K, d in ((orc.params_from_iso(0.8, 1.0, math.exp(0.5), 2), 2, 2),
                      (orc.params_from_aniso(0.7, 0.9, 1.4, 3.0), 2, 2),
                      (np.array([0.5, 0.3, 0.2, 1.0, 2.0, 5.0, 7.0, 20.0, 30.0]), 3, 2)):
        w, Th = orc.unpack_params(row, K, d)
        ll, beta = orc.loglik_general(D, y, w, Th, 0.37, mode, 100.0 ** 2)
        mll, mbeta = mp_check.loglik(D, y, w, Th, 0.37, mode, 100.0 ** 2)
        # 14 points under smooth kernels are badly conditioned (and tau^2 11' makes it worse):
        # fp64 can only be asked for cond * eps, which is also all the reference's R gets.
        S = 0.37 * np.sum(w ** 2) * orc.mixed_corr_matrix_general(D, w, Th) + (mode * 100.0 ** 2)
        tol = 20 * np.linalg.cond(S) * np.finfo(float).eps
        assert abs(ll - float(mll)) <= tol * max(1.0, abs(float(mll))) + 1e-10
        assert abs(beta - float(mbeta)) <= tol * max(1.0, abs(float(mbeta))) + 1e-12


def test_oracle_vs_mpmath_qian_and_prediction():
    D, y, Dt, _ = load_qian()
    s2 = float(np.var(y, ddof=1))
    w, Th = orc.unpack_params(orc.params_from_iso(0.8, 0.3, 15.0, 4), 2, 4)
    for mode in (0, 1):
        ll, beta = orc.loglik_general(D, y, w, Th, s2, mode, 2500.0)
        mll, mbeta = mp_check.loglik(D, y, w, Th, s2, mode, 2500.0)
        assert ll == pytest.approx(float(mll), rel=1e-10)
        assert beta == pytest.approx(float(mbeta), rel=1e-10, abs=1e-12)
    means, variances, _ = mp_check.predict(D, y, w, Th, s2, Dt[:3])
    for j in range(3):
        m, v = orc.predict_post_iso(Dt[j], D, y, 0.8, 0.3, 15.0, s2)
        assert m == pytest.approx(float(means[j]), rel=1e-10)
        assert v == pytest.approx(float(variances[j]), rel=1e-8)


def test_halton_and_qigamma_definitions():
    u = orc.runif_halton(8)
    assert u.tolist() == [0.5, 0.25, 0.75, 0.125, 0.625, 0.375, 0.875, 0.0625]
    assert float(mp_check.halton2(5)) == 0.625
    import scipy.stats as sst
    np.testing.assert_allclose(orc.qigamma(u, 7.0, 3.0), sst.invgamma.ppf(u, 7.0, scale=3.0), rtol=1e-12)


def test_golden_files_are_what_the_oracle_produces():
    """Spot re-computation of committed fixtures (guards against oracle drift)."""
    D, y, Dt, _ = load_qian()
    g = golden("hx_golden.json")
    assert len(g["grid"]["values"]) == 624
    for case in g["cases"][::5]:
        p, t1, t2 = case["draw"]
        lp = orc.logpost(D, case["theta_t"], y, case["sigma2"], "HX", (*g["theta1_pars"], *g["theta2_pars"]))
        assert lp["val"] == pytest.approx(case["val"], rel=1e-12)
        assert lp["beta"] == pytest.approx(case["beta"], rel=1e-12)
        assert orc.cond_like_log(D, y, p, t1, t2, case["sigma2"], 50.0) == pytest.approx(case["cond_like_log"], rel=1e-12)
    H = load_hyper("hx")
    i = g["grid"]["which_max"]
    m = orc.likeli_hyperpars(D, y, H[i, :2], H[i, 2:], g["grid"]["sigma2"], 1000, 50.0)
    assert math.log(m) == pytest.approx(g["grid"]["values"][i], rel=1e-11)
    mean, var, beta = orc.predict_table(D, y, g["draws"][:2], Dt, g["predict"]["sigma2"])
    np.testing.assert_allclose(mean, np.array(g["predict"]["mean"])[:2], rtol=1e-11)
    np.testing.assert_allclose(var, np.array(g["predict"]["var"])[:2], rtol=1e-9)

    gv = golden("gv_golden.json")
    for s in gv["sets"]:
        Dg, yg, Dtg, _ = load_gv(s["size"])
        mean, var, _ = orc.predict_table(Dg, yg, s["draws"][:1], Dtg[:5], s["sigma2"])
        np.testing.assert_allclose(mean[0], np.array(s["mean"])[0, :5], rtol=1e-11)

    ga = golden("adv_golden.json")
    D14 = load_maximin(14)
    assert len(ga["grid"]["values"]) == 60 and 0 <= ga["grid"]["which_max"] < 60
    c = ga["cases"][0]
    lp = orc.logpost(D14, c["theta_t"], np.array(ga["y"]), ga["sigma2"], "ADV", tuple(c["prior_pars"]))
    assert lp["val"] == pytest.approx(c["val"], rel=1e-12) and lp["like"] == pytest.approx(c["like"], rel=1e-11)


def test_config1_matern_plumbing():
    """BASELINE config 1 is CPU-only plumbing: Matern nu = 5 on an 8-point 1-D design (D1:348-374)."""
    g = golden("d1_golden.json")
    X = np.array(g["X"]).reshape(-1, 1)
    R = orc.corr_matrix_matern(g["nu"], X, 0.7)
    assert R.shape == (8, 8)
    np.testing.assert_allclose(np.diag(R), 1.0)
    np.testing.assert_allclose(R, R.T)
    assert np.all(np.linalg.eigvalsh(R) > 0)
    # closed form for half-integer nu = 5/2 cross-checks the besselK restatement
    h, th, nu = 0.37, 0.9, 2.5
    z = 2 * math.sqrt(nu) * h / th
    closed = (1 + z + z * z / 3) * math.exp(-z)
    assert float(orc.matern_corr(nu, h, th)) == pytest.approx(closed, rel=1e-12)
    for c in g["cases"]:
        lp = orc.logpost_1d(X, c["theta_t"], np.array(g["y"]), c["sigma2"], g["nu"])
        assert lp["val"] == pytest.approx(c["val"], rel=1e-12)


def test_two_family_script_restatement_and_fixture():
    """D1F: Matern + non-negative cubic spline.  Known answers of the spline (D1F:346-357), the un-normalised
    corr.vec.combined (D1F:479) and the committed fixture."""
    g = golden("d1f_golden.json")
    X = np.array(g["X"]).reshape(-1, 1)
    y = np.array(g["y"])
    assert float(orc.spline_corr(2.0, 0.0)) == 1.0
    assert float(orc.spline_corr(2.0, 1.0)) == pytest.approx(1 - 6 * 0.25 + 6 * 0.125)      # u = 1/2, first branch
    assert float(orc.spline_corr(2.0, 1.5)) == pytest.approx(2 * 0.25 ** 3)                 # u = 3/4, second branch
    assert float(orc.spline_corr(2.0, 2.5)) == 0.0
    Rs = orc.corr_matrix_spline(X, 0.45)
    assert np.all(np.linalg.eigvalsh(Rs) > 0) and np.allclose(np.diag(Rs), 1.0)
    p, t1, t2, nu = 0.7, 0.5, 0.6, g["nu"]
    r = orc.corr_vec_combined(0.41, X, p, t1, t2, nu)
    np.testing.assert_allclose(r, g["r_combined"], rtol=1e-13)
    normalised = r / (p ** 2 + (1 - p) ** 2)
    np.testing.assert_allclose(normalised, (p ** 2 * orc.corr_vec_matern(0.41, X, t1, nu) + (1 - p) ** 2 *
                                            orc.corr_vec_spline(0.41, X, t2)) / (p ** 2 + (1 - p) ** 2), rtol=1e-14)
    for c in g["cases"]:
        lp = orc.logpost_2f(X, c["theta_t"], y, c["sigma2"], nu)
        assert lp["val"] == pytest.approx(c["val"], rel=1e-12)


def test_solve_refuses_a_computationally_singular_matrix_like_base_r():
    """solve(R) inside logpost (HX:454): base R's solve.default stops when rcond < .Machine$double.eps, the reference's
    try() turns that into R.Inv <- NA.  A duplicated design point is singular only up to rounding (LU meets a pivot of
    +-1e-17, not 0): without the rcond test the restatement would return a garbage inverse where R returns NA."""
    from conftest import load_qian
    D, y, _, _ = load_qian()
    Dd = D.copy()
    Dd[40] = Dd[3]
    R = orc.mixed_corr_matrix_iso(Dd, 0.8, 0.3, 15.0)
    with pytest.raises(np.linalg.LinAlgError, match="singular"):
        orc.solve_inverse(R)
    lp = orc.logpost(Dd, [math.log(0.3), math.log(15.0), math.log(4.0)], y, 10.0, "GV")
    assert lp["R_inv"] is None and math.isnan(lp["val"])
    # a well-conditioned matrix goes through and is the LAPACK inverse
    R = orc.mixed_corr_matrix_iso(D, 0.8, 0.3, 15.0)
    np.testing.assert_array_equal(orc.solve_inverse(R), np.linalg.inv(R))


def test_log_likeli_of_the_1d_scripts_known_answer():
    """D1:437-444: log(det(R)) + n log(sigma2.MLE), against a direct evaluation through eigenvalues / lstsq."""
    g = golden("d1_golden.json")
    X, y, nu = np.array(g["X"]).reshape(-1, 1), np.array(g["y"]), g["nu"]
    R = orc.corr_matrix_matern(nu, X, 0.25)
    w = np.linalg.eigvalsh(R)
    one = np.ones(8)
    a = np.linalg.solve(R, np.stack([y, one], axis=1))
    beta = (one @ a[:, 0]) / (one @ a[:, 1])
    r = y - beta
    s2 = (r @ np.linalg.solve(R, r)) / 8
    assert orc.log_likeli_1d(nu, 0.25, X, y) == pytest.approx(np.log(w).sum() + 8 * math.log(s2), rel=1e-9)
