"""Parity of the HIP path (through the C ABI) against the oracle and the committed fixtures.

Tolerances (fp64 throughout): BASELINE.json's north_star asks for rtol 1e-8 on the Qian
predictions; likelihood terms are held to 1e-9 or tighter except where the covariance is
itself ill-conditioned (14-point designs under smooth kernels, tau^2 11' added), where
fp64 -- in R as here -- only delivers cond * eps and the tolerance says so.
"""
import math

import numpy as np
import pytest

from conftest import golden, load_gv, load_hyper, load_maximin, load_qian, synthetic_design
from oracle import ccgp_oracle as orc
from oracle import mp_check

pytestmark = pytest.mark.gpu


def digest_close(M, dig, rtol):
    scale = dig["fro"]
    assert float(M.sum()) == pytest.approx(dig["sum"], rel=rtol, abs=rtol * scale)
    assert float(np.trace(M)) == pytest.approx(dig["trace"], rel=rtol, abs=rtol * scale)
    assert float(np.linalg.norm(M)) == pytest.approx(dig["fro"], rel=rtol)
    for i, j, v in dig["entries"]:
        assert M[i, j] == pytest.approx(v, rel=rtol, abs=rtol * scale / M.shape[0])


def rand_iso_draws(rng, B, d):
    return np.stack([orc.params_from_iso(rng.uniform(0.55, 0.95), rng.uniform(0.1, 1.5), rng.uniform(8, 60), d)
                     for _ in range(B)])


# ------------------------------------------------------------------------------- a1-a5
def test_corr_kernels_qian(handle):
    from ccgp_amd.rsurface import CombinedGP
    gp = CombinedGP("HX", handle=handle)
    D, y, Dt, _ = load_qian()
    g = golden("hx_golden.json")
    th = g["corr_matrix_general"]["theta"]
    R = gp.corr_matrix(D, th)
    np.testing.assert_allclose(R, orc.corr_matrix(D, th), rtol=1e-13, atol=1e-15)
    digest_close(R, g["corr_matrix_general"]["R"], 1e-12)
    np.testing.assert_allclose(gp.corr_matrix_ISO(D, 0.45), orc.corr_matrix_iso(D, 0.45), rtol=1e-13)
    cv = g["corr_vec_iso"]
    np.testing.assert_allclose(gp.corr_vec_ISO(cv["x"], D, cv["theta"]), cv["r"], rtol=1e-12)
    np.testing.assert_allclose(gp.Mixed_corr_matrix(D, 0.8, 0.3, 15.0), orc.mixed_corr_matrix_iso(D, 0.8, 0.3, 15.0),
                               rtol=1e-13)
    np.testing.assert_allclose(gp.Mixed_corr_vec(Dt[3], D, 0.8, 0.3, 15.0),
                               orc.mixed_corr_vec_iso(Dt[3], D, 0.8, 0.3, 15.0), rtol=1e-12)
    # cross matrix with m > 64 and n not a multiple of 64 (ragged tiles)
    Dg, _, Dtg, _ = load_gv(90)
    C = handle.corr_cross(Dtg, Dg, 0.3)
    want = np.stack([orc.corr_vec_iso(x, Dg, 0.3) for x in Dtg])
    np.testing.assert_allclose(C, want, rtol=1e-12)


def test_device_exp_is_within_two_ulp_down_to_underflow(handle):
    """exp_cov (ccgp_internal.h; round 3: 256-entry table x degree-4 polynomial) in isolation.  With the new site at the
    origin the expanded distance of corr.vec (HX:373) is (0 - 2 * 0) + theta x_i^2, i.e. the correctly rounded product
    itself, so the entry is exp(-fl(fl(x^2) theta)) and can be compared with a 50-digit exp of that very double:
    <= 2 ulp over 20 000 arguments in [-745, 0] (gradual underflow included), exactly 1 at 0, exactly 0 beyond the
    underflow threshold, NaN for a NaN coordinate."""
    import mpmath as mp
    mp.mp.dps = 50
    rng = np.random.default_rng(7)
    x = np.sqrt(np.concatenate([rng.uniform(0, 1, 6000), rng.uniform(0, 40, 8000), rng.uniform(0, 745, 6000),
                                [0.0, 750.0, 1000.0, 5000.0, 1e8]]))
    theta = 1.0000001
    r = handle.corr_cross(np.zeros((1, 1)), x[:, None], [theta])[0]
    arg = -((x * x) * theta)
    want = np.array([float(mp.exp(mp.mpf(float(a)))) for a in arg])
    assert r[-5] == 1.0 and (r[-4:] == 0.0).all()
    ok = want > 2.3e-308                       # normal range: relative ulp bound
    assert np.max(np.abs(r[ok] - want[ok]) / np.spacing(want[ok])) <= 2.0
    assert np.max(np.abs(r[~ok] - want[~ok])) <= 2 * 4.95e-324 * 2 ** 1    # denormals: within a couple of quanta
    xn = x[:64].copy()
    xn[5] = np.nan
    rn = handle.corr_cross(np.zeros((1, 1)), xn[:, None], [theta])[0]
    assert np.isnan(rn[5]) and np.isfinite(np.delete(rn, 5)).all()


def test_small_n_evaluation_does_not_depend_on_the_batch_it_travels_in(handle):
    """Batches of <= 64 evaluations run on the 16 x 16 thread grid (four waves per matrix: latency), larger ones at
    n <= 64 on the 8 x 8 grid (one wave per matrix: throughput).  Every matrix entry sees the same rank-1 updates in
    the same order either way, so an evaluation's bits must not depend on which one it got."""
    D, y, _, _ = load_qian()
    rng = np.random.default_rng(21)
    P = np.array([orc.params_from_iso(rng.uniform(0.5, 0.95), rng.uniform(0.2, 1.0), rng.uniform(5, 30), 4) for _ in range(150)])
    for mode, tau2 in ((0, 0.0), (1, 2500.0)):
        big = handle.loglik_batch(D, y, 2, P, 37.0, mode, tau2)
        for lo, hi in ((0, 1), (7, 8), (100, 140)):
            small = handle.loglik_batch(D, y, 2, P[lo:hi], 37.0, mode, tau2)
            np.testing.assert_array_equal(small[0], big[0][lo:hi])
            np.testing.assert_array_equal(small[1], big[1][lo:hi])
    # a 50-point design (n < 64, not a multiple of 8 or 16)
    Dg, yg, _, _ = load_gv(50)
    Pg = np.array([orc.params_from_iso(0.7, 0.3 + 0.01 * i, 15.0, 9) for i in range(90)])
    big = handle.loglik_batch(Dg, yg, 2, Pg, 10.0)
    np.testing.assert_array_equal(handle.loglik_batch(Dg, yg, 2, Pg[40:43], 10.0)[0], big[0][40:43])


def test_corr_kernels_aniso(handle):
    from ccgp_amd.rsurface import CombinedGP
    gp = CombinedGP("ANI", handle=handle)
    D = load_maximin(100)
    np.testing.assert_allclose(gp.corr_matrix(D, 0.8, 1.5), orc.corr_matrix(D, [0.8, 1.5]), rtol=1e-13)
    np.testing.assert_allclose(gp.Mixed_corr_matrix(D, 0.7, 0.8, 1.5, 6.0),
                               orc.mixed_corr_matrix_aniso(D, 0.7, 0.8, 1.5, 6.0), rtol=1e-13)
    np.testing.assert_allclose(gp.corr_vec([0.1, -0.4], D, 0.8, 1.5), orc.corr_vec([0.1, -0.4], D, [0.8, 1.5]), rtol=1e-12)
    np.testing.assert_allclose(gp.Mixed_corr_vec([0.1, -0.4], D, 0.7, 0.8, 1.5, 6.0),
                               orc.mixed_corr_vec_aniso([0.1, -0.4], D, 0.7, 0.8, 1.5, 6.0), rtol=1e-12)


# ------------------------------------------------------------------------------- a6, a7, a10, a11 literal forms
def test_rinv_helpers_and_literal_predict_post(handle):
    from ccgp_amd.rsurface import CombinedGP
    gp = CombinedGP("HX", handle=handle)
    D, y, Dt, _ = load_qian()
    s2 = 10.0
    R_inv = orc.solve_inverse(orc.mixed_corr_matrix_iso(D, 0.8, 0.3, 15.0))
    beta = orc.beta_mle(R_inv, y)
    assert gp.beta_MLE(R_inv, y) == pytest.approx(beta, rel=1e-11)
    assert gp.sigma2_MLE(R_inv, y, beta) == pytest.approx(orc.sigma2_mle(R_inv, y, beta), rel=1e-9)
    mf, v1, v2 = orc.factors(R_inv, beta, y)
    f = gp.factors(np.concatenate([R_inv.ravel(order="F"), [beta]]), 64, y)
    np.testing.assert_allclose(f[:64], mf, rtol=1e-9, atol=1e-9 * np.abs(mf).max())
    np.testing.assert_allclose(f[64:128], v1, rtol=1e-9, atol=1e-9 * np.abs(v1).max())
    assert f[128] == pytest.approx(v2, rel=1e-9)
    pars = np.concatenate([[0.8, 0.3, 15.0, beta], mf, v1, [v2], R_inv.ravel(order="F")])
    for j in (0, 5, 13):
        got = gp.predict_post(Dt[j], D, pars, s2)
        want = orc.predict_post_iso(Dt[j], D, y, 0.8, 0.3, 15.0, s2)
        assert got[0, 0] == pytest.approx(want[0], rel=1e-10)
        assert got[0, 1] == pytest.approx(want[1], rel=1e-8, abs=1e-10 * s2)


def test_rinv_helpers_take_an_asymmetric_inverse_as_the_reference_does(handle):
    """R's solve() output is symmetric only up to rounding and the helpers accept ANY caller-supplied R.Inv:
    var.factor1 = apply(R.Inv, 2, sum) are column sums (HX:609), beta.MLE's numerator is (1' R.Inv) y (HX:387),
    mean.factor = R.Inv %*% (y - beta) row products (HX:608).  A visibly asymmetric matrix tells the two apart."""
    from ccgp_amd.rsurface import CombinedGP
    gp = CombinedGP("HX", handle=handle)
    rng = np.random.default_rng(77)
    for n in (7, 64, 90, 200):
        A = rng.normal(size=(n, n)) + n * np.eye(n)          # nothing symmetric about it
        y = rng.normal(size=n)
        beta = orc.beta_mle(A, y)
        assert abs(beta - float((A.sum(axis=1) @ y) / A.sum())) > 1e-6 * abs(beta)    # row sums WOULD differ
        assert gp.beta_MLE(A, y) == pytest.approx(beta, rel=1e-11)
        assert gp.sigma2_MLE(A, y, beta) == pytest.approx(orc.sigma2_mle(A, y, beta), rel=1e-10)
        mf, v1, v2 = orc.factors(A, beta, y)
        f = gp.factors(np.concatenate([A.ravel(order="F"), [beta]]), n, y)
        np.testing.assert_allclose(f[:n], mf, rtol=1e-11, atol=1e-12 * np.abs(mf).max())
        np.testing.assert_allclose(f[n:2 * n], v1, rtol=1e-12)
        assert f[2 * n] == pytest.approx(v2, rel=1e-12)


# ------------------------------------------------------------------------------- a8 / a12, small path
@pytest.mark.parametrize("mode", [0, 1])
def test_loglik_batch_qian_vs_oracle(handle, mode):
    D, y, _, _ = load_qian()
    s2 = float(np.var(y, ddof=1))
    rng = np.random.default_rng(11)
    P = rand_iso_draws(rng, 24, 4)
    ll, beta, st = handle.loglik_batch(D, y, 2, P, s2, mode, 2500.0)
    assert not st.any()
    for b in range(P.shape[0]):
        w, Th = orc.unpack_params(P[b], 2, 4)
        wl, wb = orc.loglik_general(D, y, w, Th, s2, mode, 2500.0)
        assert ll[b] == pytest.approx(wl, rel=1e-10)
        assert beta[b] == pytest.approx(wb, rel=1e-10, abs=1e-13)


@pytest.mark.parametrize("n", [5, 8, 9, 31, 48, 63, 64, 65, 80, 97, 112, 127, 128])
def test_fused_evaluator_size_sweep(handle, n):
    """Every template instance of the register-resident evaluator (G = 8: n <= 64, G = 16: n <= 128),
    K = 1, 2 and 3 components, both likelihood modes."""
    X, y = synthetic_design(n, 3, seed=100 + n)
    rng = np.random.default_rng(n)
    for K in (1, 2, 3):
        B = 3
        P = np.empty((B, K + 3 * K))
        for b in range(B):
            w = rng.dirichlet(np.ones(K)) if K > 1 else np.array([1.0])
            th = np.exp(rng.uniform(np.log(2.0), np.log(60.0), size=(K, 3)))
            th[-1] = np.maximum(th[-1], 30.0)
            P[b] = np.concatenate([w, th.ravel()])
        for mode in (0, 1):
            ll, beta, st = handle.loglik_batch(X, y, K, P, 0.9, mode, 25.0)
            assert not st.any()
            for b in range(B):
                w, Th = orc.unpack_params(P[b], K, 3)
                wl, wb = orc.loglik_general(X, y, w, Th, 0.9, mode, 25.0)
                assert ll[b] == pytest.approx(wl, rel=1e-9, abs=1e-9)
                assert beta[b] == pytest.approx(wb, rel=1e-8, abs=1e-11)


def test_large_input_dimension_routes_to_a_path_that_fits(handle):
    """d = 40 at n = 120 does not fit the LDS budget of the fused kernels' design copy for every
    variant; whichever path runs, the answer is the oracle's."""
    rng = np.random.default_rng(5)
    X = rng.random((120, 40))
    y = np.sin(X.sum(axis=1))
    P = np.concatenate([[0.6, 0.4], np.full(40, 0.05), np.full(40, 2.0)])[None]
    ll, beta, st = handle.loglik_batch(X, y, 2, P, 1.0, 0, 0.0)
    w, Th = orc.unpack_params(P[0], 2, 40)
    wl, wb = orc.loglik_general(X, y, w, Th, 1.0)
    assert st[0] == 0 and ll[0] == pytest.approx(wl, rel=1e-9) and beta[0] == pytest.approx(wb, rel=1e-8)
    X2 = rng.random((128, 64))
    P2 = np.concatenate([[0.6, 0.4], np.full(64, 0.03), np.full(64, 1.0)])[None]
    y2 = np.cos(X2.sum(axis=1))
    ll2, _, st2 = handle.loglik_batch(X2, y2, 2, P2, 1.0, 1, 4.0)
    w2, Th2 = orc.unpack_params(P2[0], 2, 64)
    assert st2[0] == 0 and ll2[0] == pytest.approx(orc.loglik_general(X2, y2, w2, Th2, 1.0, 1, 4.0)[0], rel=1e-9)


def test_logpost_golden_every_script(handle):
    from ccgp_amd.rsurface import CombinedGP
    D, y, _, _ = load_qian()
    g = golden("hx_golden.json")
    gp = CombinedGP("HX", handle=handle)
    for c in g["cases"]:
        r = gp.logpost(D, c["theta_t"], y, c["sigma2"], g["theta1_pars"], g["theta2_pars"])
        assert r["val"] == pytest.approx(c["val"], rel=1e-10)
        assert r["beta"] == pytest.approx(c["beta"], rel=1e-10)
        digest_close(r["R_Inv"], c["R_inv"], 1e-8)

    ga = golden("ani_golden.json")
    D100, y100 = load_maximin(100), np.array(ga["y"])
    gpa = CombinedGP("ANI", handle=handle)
    for c in ga["cases"]:
        r = gpa.logpost(D100, c["theta_t"], y100, c["sigma2"])
        # 100 points under these smooth draws: cond(R) = 4e5 .. 1.4e8
        assert r["val"] == pytest.approx(c["val"], rel=1e-7)
        assert r["beta"] == pytest.approx(c["beta"], rel=1e-6, abs=1e-9)
        digest_close(r["R_Inv"], c["R_inv"], 1e-5)
    # the loose bounds above are cond * eps, not slack: arbitrate the likelihood term of the best- and the
    # worst-conditioned case at 50 digits (oracle/mp_check.py) -- device and oracle must both sit within cond * eps of it
    def ani_draw(tt):
        p, t1, t2, lam = 1.0 / (1.0 + math.exp(-tt[2])), math.exp(tt[0]), math.exp(tt[1]), math.exp(tt[3])
        return orc.unpack_params(np.array([p, 1 - p, t1, t2, (1 + lam) * t1, (1 + lam) * t2]), 2, 2)
    conds = [np.linalg.cond(orc.mixed_corr_matrix_general(D100, *ani_draw(c["theta_t"]))) for c in ga["cases"]]
    for c in (ga["cases"][int(np.argmin(conds))], ga["cases"][int(np.argmax(conds))]):
        tt = c["theta_t"]
        w, Th = ani_draw(tt)
        truth_ll, truth_beta = (float(v) for v in mp_check.loglik(D100, y100, w, Th, c["sigma2"], 0, 0.0))
        cond = np.linalg.cond(orc.mixed_corr_matrix_general(D100, w, Th))
        tol = 50 * cond * np.finfo(float).eps
        r = gpa.logpost(D100, tt, y100, c["sigma2"])
        prior_jac = orc.logpost(D100, tt, y100, c["sigma2"], "ANI")
        extra = prior_jac["val"] - prior_jac["log_like"]            # log-Jacobian + log-prior (ANI:457-462): closed forms
        assert abs((r["val"] - extra) - truth_ll) <= tol * abs(truth_ll) + 1e-9
        assert abs(c["loglik"] - truth_ll) <= tol * abs(truth_ll) + 1e-9
        assert abs(r["beta"] - truth_beta) <= tol * max(abs(truth_beta), 1.0)

    gv = golden("gv_golden.json")
    gpg = CombinedGP("GV", handle=handle)
    for s in gv["sets"]:
        Dg, yg, _, _ = load_gv(s["size"])
        for c in s["cases"]:
            r = gpg.logpost(Dg, c["theta_t"], yg, s["sigma2"])
            assert r["val"] == pytest.approx(c["val"], rel=1e-10)
            assert r["beta"] == pytest.approx(c["beta"], rel=1e-9, abs=1e-12)
            digest_close(r["R_Inv"], c["R_inv"], 1e-8)

    gd = golden("adv_golden.json")
    D14, y14 = load_maximin(14), np.array(gd["y"])
    gpd = CombinedGP("ADV", handle=handle)
    for c in gd["cases"]:
        r = gpd.logpost(D14, c["theta_t"], y14, gd["sigma2"], c["prior_pars"][:2], c["prior_pars"][2:])
        # 14 smooth points: cond(R) ~ 1e8..1e11, so only cond*eps can be asked for
        assert r["val"] == pytest.approx(c["val"], rel=1e-6)
        assert r["like"] == pytest.approx(c["like"], rel=1e-4)

    # ISO / BSQ share one prior (ISO:453 = BSQ:450)
    t = [0.1, 2.5, 0.9]
    for script in ("ISO", "BSQ"):
        r = CombinedGP(script, handle=handle).logpost(D, t, y, 10.0)
        assert r["val"] == pytest.approx(orc.logpost(D, t, y, 10.0, script)["val"], rel=1e-10)


# ------------------------------------------------------------------------------- a9
def test_hx_grid_full_624_rows(handle):
    """BASELINE config 2: 624 rows x 1000 Halton nodes on the Qian design (HX:584-595)."""
    from ccgp_amd.rsurface import CombinedGP
    D, y, _, _ = load_qian()
    g = golden("hx_golden.json")["grid"]
    H = load_hyper("hx")
    vals, arg, logs = handle.grid_marginal(D, y, g["sigma2"], H, g["N"], g["tau"], g["take_log"], want_logs=True)
    np.testing.assert_allclose(logs[0], g["row0_logs"], rtol=1e-9)
    np.testing.assert_allclose(vals, g["values"], rtol=1e-9)
    assert arg == g["which_max"]
    r = CombinedGP("HX", handle=handle).choose_hyperpars(D, y, H[:16], g["sigma2"])
    np.testing.assert_allclose(r["likelihoods"], g["values"][:16], rtol=1e-9)
    np.testing.assert_array_equal(r["pars"], H[int(np.argmax(g["values"][:16]))])
    one = CombinedGP("HX", handle=handle).likeli_hyperpars(D, y, H[7, :2], H[7, 2:], g["sigma2"])
    assert math.log(one) == pytest.approx(g["values"][7], rel=1e-9)


def test_config3_aniso_grid(handle):
    """BASELINE config 3: 60 x 1728 grid on maximin-100 with the anisotropic kernel."""
    ga = golden("ani_golden.json")
    g = ga["grid"]
    D, y = load_maximin(100), np.array(ga["y"])
    # scale columns x16: as bundled (tuned for 14 points) cond(Sigma) reaches 7e15 on this design
    H = load_hyper("adv") * np.array([1.0, g["b_scale"], 1.0, g["b_scale"]])
    vals, arg, logs = handle.grid_marginal(D, y, g["sigma2"], H, g["N"], g["tau"], g["take_log"],
                                           aniso_lambda=g["aniso_lambda"], want_logs=True)
    np.testing.assert_allclose(logs[0], g["row0_logs"], rtol=1e-8, atol=1e-7)     # cond <= ~1e8
    np.testing.assert_allclose(vals, g["values"], rtol=1e-6)
    assert arg == g["which_max"]


def test_adv_grid(handle):
    from ccgp_amd.rsurface import CombinedGP
    gd = golden("adv_golden.json")
    D14, y14 = load_maximin(14), np.array(gd["y"])
    H = load_hyper("adv")
    r = CombinedGP("ADV", handle=handle).choose_hyperpars(D14, y14, H, gd["sigma2"])
    np.testing.assert_allclose(r["likelihoods"], gd["grid"]["values"], rtol=1e-4)
    assert r["which_max"] == gd["grid"]["which_max"]
    # arbitrate two nodes at 50 digits: the HIP value must be within cond*eps of the truth
    u = orc.runif_halton(4)
    for j in (0, 3):
        th1, th2 = orc.qigamma(u[j], H[5, 0], H[5, 1]), orc.qigamma(u[j], H[5, 2], H[5, 3])
        row = orc.params_from_iso(u[j], float(th1), float(th2), 2)
        w, Th = orc.unpack_params(row, 2, 2)
        got, _, _ = handle.loglik_batch(D14, y14, 2, row[None], gd["sigma2"], 1, 100.0 ** 2)
        S = gd["sigma2"] * np.sum(w ** 2) * orc.mixed_corr_matrix_general(D14, w, Th) + 100.0 ** 2
        tol = 50 * np.linalg.cond(S) * np.finfo(float).eps
        truth = float(mp_check.loglik(D14, y14, w, Th, gd["sigma2"], 1, 100.0 ** 2)[0])
        assert abs(got[0] - truth) <= tol * abs(truth) + 1e-9


# ------------------------------------------------------------------------------- a10 + a11 batched
def test_qian_predictions_rtol_1e8(handle):
    """north_star: predictions match the reference math to rtol 1e-8 on the Qian set."""
    from ccgp_amd.rsurface import CombinedGP
    D, y, Dt, _ = load_qian()
    g = golden("hx_golden.json")
    t = CombinedGP("HX", handle=handle).prediction_table(Dt, g["draws"], D, g["predict"]["sigma2"], y)
    assert not t["status"].any()
    np.testing.assert_allclose(t["mean"], g["predict"]["mean"], rtol=1e-8)
    np.testing.assert_allclose(t["var"], g["predict"]["var"], rtol=1e-8)
    np.testing.assert_allclose(t["beta"], g["predict"]["beta"], rtol=1e-8)
    np.testing.assert_allclose(t["y_hat"], np.mean(g["predict"]["mean"], axis=0), rtol=1e-8)


def test_ground_vibrations_prediction_tables(handle):
    """BASELINE config 5 shapes: n = 50 / 90, d = 9, m = 150 / 110 (chunked test points)."""
    from ccgp_amd.rsurface import CombinedGP
    gv = golden("gv_golden.json")
    gp = CombinedGP("GV", handle=handle)
    for s in gv["sets"]:
        Dg, yg, Dtg, _ = load_gv(s["size"])
        t = gp.prediction_table(Dtg, s["draws"], Dg, s["sigma2"], yg)
        np.testing.assert_allclose(t["mean"], s["mean"], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(t["var"], s["var"], rtol=1e-8, atol=1e-10 * s["sigma2"])
        np.testing.assert_allclose(t["beta"], s["beta"], rtol=1e-8, atol=1e-11)


def test_aniso_prediction_table(handle):
    from ccgp_amd.rsurface import CombinedGP
    ga = golden("ani_golden.json")
    D, y = load_maximin(100), np.array(ga["y"])
    t = CombinedGP("ANI", handle=handle).prediction_table(np.array(ga["Xtest"]), ga["draws"], D,
                                                         ga["predict"]["sigma2"], y)
    # cond(R) up to 1.4e8 for these draws (see test_logpost_golden_every_script)
    np.testing.assert_allclose(t["mean"], ga["predict"]["mean"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(t["var"], ga["predict"]["var"], rtol=1e-5, atol=1e-7 * ga["predict"]["sigma2"])


def test_factors_frame_converter(handle):
    from ccgp_amd.rsurface import CombinedGP
    D, y, Dt, _ = load_qian()
    gp = CombinedGP("HX", handle=handle)
    frame = gp.factors_frame_from_draws([(0.8, 0.3, 15.0), (0.7, 0.5, 25.0)], D, 10.0, y)
    assert frame.shape == (2, 5 + 2 * 64 + 64 * 64)            # HX:631-643 row width
    got = gp.predict_post(Dt[2], D, frame[1], 10.0)
    want = orc.predict_post_iso(Dt[2], D, y, 0.7, 0.5, 25.0, 10.0)
    assert got[0, 0] == pytest.approx(want[0], rel=1e-9) and got[0, 1] == pytest.approx(want[1], rel=1e-7)


# ------------------------------------------------------------------------------- blocked path (n > 128)
@pytest.mark.parametrize("n,d,K", [(129, 3, 2), (200, 5, 3), (256, 5, 3), (640, 5, 3), (1000, 4, 2)])
@pytest.mark.parametrize("mode", [0, 1])
def test_blocked_path_vs_oracle(handle, n, d, K, mode):
    X, y = synthetic_design(n, d, seed=n)
    rng = np.random.default_rng(n + mode)
    B = 5
    P = np.empty((B, K + K * d))
    for b in range(B):
        w = rng.dirichlet(np.ones(K))
        th = np.exp(rng.uniform(np.log(0.5), np.log(30.0), size=(K, d)))
        th[-1] = np.maximum(th[-1], 40.0)     # the rough component keeps R numerically PD
        P[b] = np.concatenate([w, th.ravel()])
    ll, beta, st = handle.loglik_batch(X, y, K, P, 1.3, mode, 9.0)
    assert not st.any()
    for b in range(B):
        w, Th = orc.unpack_params(P[b], K, d)
        wl, wb = orc.loglik_general(X, y, w, Th, 1.3, mode, 9.0)
        assert ll[b] == pytest.approx(wl, rel=1e-9)
        assert beta[b] == pytest.approx(wb, rel=1e-8, abs=1e-11)


@pytest.mark.parametrize("n,m", [(300, 150), (200, 1), (520, 260)])
def test_blocked_path_prediction(handle, n, m):
    """a10 + a11 for n > 128: the cross-correlation rows ride along as extra tile rows."""
    X, y = synthetic_design(n, 3, seed=n + 1)
    Xt = np.random.default_rng(n).random((m, 3))
    draws = [(0.7, 3.0, 40.0), (0.6, 6.0, 90.0), (0.8, 2.0, 60.0)]
    P = np.stack([orc.params_from_iso(p, t1, t2, 3) for p, t1, t2 in draws])
    mean, var, beta, st = handle.predict_batch(X, y, 2, P, Xt, 1.7)
    assert not st.any()
    wm, wv, wb = orc.predict_table(X, y, draws, Xt, 1.7)
    np.testing.assert_allclose(beta, wb, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(mean, wm, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(var, wv, rtol=1e-7, atol=1e-9 * 1.7)


def test_blocked_path_gradient_and_inverse(handle):
    """n > 128: identity rows ride along as extra tile rows, R^-1 tiles are formed in registers
    and contracted with the kernel derivatives (gradient) or written out (solve(R), HX:454)."""
    n, d, K = 300, 3, 2
    X, y = synthetic_design(n, d, seed=77)
    rows = np.stack([np.concatenate([[0.7, 0.3], [3.0, 4.0, 5.0], [40.0, 50.0, 60.0]]),
                     np.concatenate([[0.5, 0.5], [6.0, 2.0, 9.0], [80.0, 30.0, 45.0]])])
    ll, beta, grad, st = handle.loglik_grad_batch(X, y, K, rows, 1.3)
    assert not st.any()
    for b in range(2):
        w, Th = orc.unpack_params(rows[b], K, d)
        wl, wb = orc.loglik_general(X, y, w, Th, 1.3)
        assert ll[b] == pytest.approx(wl, rel=1e-9) and beta[b] == pytest.approx(wb, rel=1e-8, abs=1e-11)
        fd = orc.loglik_grad_fd(X, y, rows[b], K, d, 1.3)
        np.testing.assert_allclose(grad[b], fd, rtol=2e-5, atol=2e-5 * np.abs(fd).max())
    # explicit inverse through logpost (isotropic GV script), n = 300 and a ragged n = 257
    from ccgp_amd.rsurface import CombinedGP
    gp = CombinedGP("GV", handle=handle)
    for nn in (300, 257):
        t = [math.log(4.0), math.log(50.0), 0.8]
        r = gp.logpost(X[:nn], t, y[:nn], 1.3)
        want = orc.logpost(X[:nn], t, y[:nn], 1.3, "GV")
        assert r["val"] == pytest.approx(want["val"], rel=1e-9)
        np.testing.assert_allclose(r["R_Inv"], want["R_inv"], rtol=1e-7, atol=1e-8 * np.abs(want["R_inv"]).max())


@pytest.mark.parametrize("n,d,K", [(200, 8, 2), (150, 11, 3), (260, 1, 4), (300, 6, 1), (140, 20, 2)])
def test_blocked_gradient_contraction_instances(handle, n, d, K):
    """grad_contract_kernel is instantiated for d <= 4, 6, 8 (straight-line contraction, theta = 0 on the padded dimensions)
    and for any d (groups of eight dimensions, the exp recomputed per group); one component per pass, so any K.  Every
    instance against central differences of the device log-likelihood in every coordinate, ragged n (padding rows of M are
    zero), and a failing draw (NaN gradient, status set)."""
    X, y = synthetic_design(n, d, seed=11 * n + d)
    rng = np.random.default_rng(n + d + K)
    rough = 2.0 * n ** (2.0 / d) / d
    rows = np.empty((3, K + K * d))
    for b in range(3):
        th = np.exp(rng.uniform(np.log(0.02 * rough), np.log(0.3 * rough), size=(K, d)))
        th[-1] = rng.uniform(rough, 2.0 * rough, d)
        rows[b] = np.concatenate([0.2 + 0.6 * rng.dirichlet(np.ones(K)), th.ravel()])
    rows[2, K:] = 0.0                                  # R = 11': fails
    ll, beta, grad, st = handle.loglik_grad_batch(X, y, K, rows, 1.3)
    assert st[0] == 0 and st[1] == 0 and st[2] != 0 and np.isnan(grad[2]).all() and np.isfinite(grad[:2]).all()
    ll0, _, _ = handle.loglik_batch(X, y, K, rows, 1.3)
    np.testing.assert_array_equal(ll[:2], ll0[:2])     # the sweep with identity rows gives the same likelihood bits
    P = rows.shape[1]
    for b in range(2):
        pert = np.repeat(rows[b][None], 2 * P, axis=0)
        hstep = 1e-5 * np.abs(rows[b])
        for j in range(P):
            pert[2 * j, j] += hstep[j]
            pert[2 * j + 1, j] -= hstep[j]
        llp, _, stp = handle.loglik_batch(X, y, K, pert, 1.3)
        assert not stp.any()
        fd = (llp[0::2] - llp[1::2]) / (2 * hstep)
        np.testing.assert_allclose(grad[b], fd, rtol=2e-4, atol=2e-4 * np.abs(fd).max(), err_msg="n=%d d=%d K=%d" % (n, d, K))


def test_gradient_at_n4096_matches_central_differences_of_the_device_likelihood(handle):
    """Full BASELINE config 4 size: the analytic gradient (identity rows + R^-1 contraction)
    against central differences of the device log-likelihood itself, every coordinate."""
    import time
    n, d, K = 4096, 5, 3
    X, y = synthetic_design(n, d, seed=20140101)
    rng = np.random.default_rng(3)
    w = 0.15 + 0.55 * rng.dirichlet(np.ones(K))
    th = np.exp(rng.uniform(np.log(0.5), np.log(50.0), size=(K, d)))
    th[-1] = np.maximum(th[-1], 20.0)
    row = np.concatenate([w, th.ravel()])
    t0 = time.perf_counter()
    ll, beta, grad, st = handle.loglik_grad_batch(X, y, K, row[None], 1.0)
    t_grad = time.perf_counter() - t0
    assert st[0] == 0 and np.all(np.isfinite(grad))
    P = row.size
    pert = np.repeat(row[None], 2 * P, axis=0)
    hstep = 1e-5 * np.maximum(1.0, np.abs(row))
    for j in range(P):
        pert[2 * j, j] += hstep[j]
        pert[2 * j + 1, j] -= hstep[j]
    llp, _, stp = handle.loglik_batch(X, y, K, pert, 1.0)
    assert not stp.any()
    fd = (llp[0::2] - llp[1::2]) / (2 * hstep)
    np.testing.assert_allclose(grad[0], fd, rtol=5e-4, atol=5e-4 * np.abs(fd).max())
    print("n=4096 gradient (1 draw, host API incl. allocation): %.1f ms" % (1e3 * t_grad))


@pytest.mark.parametrize("n", [300, 1100])
def test_update_loops_with_buffer_offsets_and_with_pointers_give_the_same_bits(handle, n):
    """The round-3 update loops (whole tiles, diagonal workgroup) address their panels through 32-bit buffer offsets and
    fall back to the 64-bit-pointer loops of round 2 when a panel spans 4 GiB or more.  OPT_WIDE_OFFSETS forces the
    fallback: log-likelihood, prediction (extra tile rows) and gradient (identity rows) must agree bit for bit."""
    from ccgp_amd import api
    d, K = 4, 2
    X, y = synthetic_design(n, d, seed=3 * n)
    rng = np.random.default_rng(n)
    B = 9                                     # not a multiple of 8: ragged matrix groups
    P = np.empty((B, K + K * d))
    for b in range(B):
        w = rng.dirichlet(np.ones(K))
        th = np.exp(rng.uniform(np.log(0.5), np.log(30.0), size=(K, d)))
        th[-1] = np.maximum(th[-1], 40.0)
        P[b] = np.concatenate([w, th.ravel()])
    Xt = rng.random((130, d))
    def run():
        ll, beta, st = handle.loglik_batch(X, y, K, P, 1.3)
        mean, var, _, st2 = handle.predict_batch(X, y, K, P[:3], Xt, 1.3)
        _, _, grad, st3 = handle.loglik_grad_batch(X, y, K, P[:2], 1.3)
        assert not st.any() and not st2.any() and not st3.any()
        return ll, beta, mean, var, grad
    got = run()
    handle.set_option(api.OPT_WIDE_OFFSETS, 1)
    try:
        want = run()
    finally:
        handle.set_option(api.OPT_WIDE_OFFSETS, 0)
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a, b)
    w, Th = orc.unpack_params(P[0], K, d)
    assert got[0][0] == pytest.approx(orc.loglik_general(X, y, w, Th, 1.3)[0], rel=1e-9)


@pytest.mark.parametrize("n,B", [(1100, 64), (1100, 16), (700, 9), (1537, 40)])
def test_tail_tiles_whole_as_half_strips_and_as_quarter_strips_give_the_same_bits(handle, n, B):
    """The tiles of an update launch's last, partial step of 256 workgroups run whole (OPT_TAIL_STRIPS 0), as two half-width
    ring strips (2: rounds 2 - 3) or as four quarter-width ones where that fills the step (1: default since round 4).  With 64
    matrices at n = 1100 the launches of block columns 2, 3, 4 and 7 end in three quarters, a half, a quarter of a step
    and a single partial step: every policy is exercised.  Each output element sums its k four at a time in ascending order
    in all three forms: likelihood, prediction (extra tile rows) and gradient (identity rows) bit for bit."""
    from ccgp_amd import api
    d, K = 3, 2
    X, y = synthetic_design(n, d, seed=3 * n + B)
    rng = np.random.default_rng(n + B)
    P = np.empty((B, K + K * d))
    for b in range(B):
        th = np.exp(rng.uniform(np.log(0.5), np.log(30.0), size=(K, d)))
        th[-1] = np.maximum(th[-1], 2.0 * n ** (2.0 / d) / d)
        P[b] = np.concatenate([rng.dirichlet(np.ones(K)), th.ravel()])
    Xt = rng.random((70, d))
    def run(mode):
        handle.set_option(api.OPT_TAIL_STRIPS, mode)
        try:
            ll, beta, st = handle.loglik_batch(X, y, K, P, 1.3)
            mean, var, _, st2 = handle.predict_batch(X, y, K, P[:5], Xt, 1.3)
            _, _, grad, st3 = handle.loglik_grad_batch(X, y, K, P[:3], 1.3)
        finally:
            handle.set_option(api.OPT_TAIL_STRIPS, 1)
        assert not st.any() and not st2.any() and not st3.any()
        return ll, beta, mean, var, grad
    whole, quarters, halves = run(0), run(1), run(2)
    for a, b, c in zip(whole, quarters, halves):
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(a, c)
    w, Th = orc.unpack_params(P[0], K, d)
    assert whole[0][0] == pytest.approx(orc.loglik_general(X, y, w, Th, 1.3)[0], rel=1e-8)


def test_small_and_blocked_agree_across_the_cutover(handle):
    """n = 128 runs the fused kernel, n = 129 the blocked one: appending one far-away,
    nearly independent point must change the likelihood by exactly its own marginal term."""
    X, y = synthetic_design(129, 2, seed=5)
    P = orc.params_from_iso(0.7, 30.0, 200.0, 2)[None]
    a, _, _ = handle.loglik_batch(X[:128], y[:128], 2, P, 1.0, 1, 4.0)
    b, _, _ = handle.loglik_batch(X, y, 2, P, 1.0, 1, 4.0)
    w, Th = orc.unpack_params(P[0], 2, 2)
    assert a[0] == pytest.approx(orc.loglik_general(X[:128], y[:128], w, Th, 1.0, 1, 4.0)[0], rel=1e-10)
    assert b[0] == pytest.approx(orc.loglik_general(X, y, w, Th, 1.0, 1, 4.0)[0], rel=1e-10)


def test_n4096_against_lapack_and_invariances(handle):
    """BASELINE config 4 size.  One direct LAPACK comparison (seconds on CPU) plus
    size-independent properties: determinism, batch-position independence, invariance of the
    likelihood under a permutation of the design rows."""
    n, d, K = 4096, 5, 3
    X, y = synthetic_design(n, d, seed=20140101)
    rng = np.random.default_rng(7)
    B = 8
    P = np.empty((B, K + K * d))
    for b in range(B):
        w = rng.dirichlet(np.ones(K))
        th = np.exp(rng.uniform(np.log(0.5), np.log(50.0), size=(K, d)))
        th[-1] = np.maximum(th[-1], 20.0)
        P[b] = np.concatenate([w, th.ravel()])
    P[5] = P[2]                                   # same draw at two batch positions
    ll, beta, st = handle.loglik_batch(X, y, K, P, 1.0, 0, 0.0)
    assert not st.any() and np.all(np.isfinite(ll))
    assert ll[5] == ll[2] and beta[5] == beta[2]
    ll2, beta2, _ = handle.loglik_batch(X, y, K, P, 1.0, 0, 0.0)
    np.testing.assert_array_equal(ll, ll2)        # deterministic
    perm = np.random.default_rng(1).permutation(n)
    llp, betap, _ = handle.loglik_batch(X[perm], y[perm], K, P[:2], 1.0, 0, 0.0)
    np.testing.assert_allclose(llp, ll[:2], rtol=1e-9)
    np.testing.assert_allclose(betap, beta[:2], rtol=1e-7, atol=1e-10)
    # direct check of one draw with LAPACK Cholesky (not the LU inverse: seconds, not minutes)
    import scipy.linalg as sla
    w, Th = orc.unpack_params(P[0], K, d)
    R = orc.mixed_corr_matrix_general(X, w, Th)
    L = sla.cholesky(R, lower=True)
    zy = sla.solve_triangular(L, y, lower=True)
    z1 = sla.solve_triangular(L, np.ones(n), lower=True)
    b0 = (z1 @ zy) / (z1 @ z1)
    c = 1.0 * np.sum(w ** 2)
    want = -0.5 * (n * math.log(2 * math.pi) + n * math.log(c) + 2 * np.log(np.diag(L)).sum()
                   + np.sum((zy - b0 * z1) ** 2) / c)
    assert ll[0] == pytest.approx(want, rel=1e-9)
    assert beta[0] == pytest.approx(b0, rel=1e-7, abs=1e-10)


def test_beyond_the_benchmark_size_n8192_ragged(handle):
    """Twice config 4's order and not a multiple of the tile: n = 8100 (64 block columns, ragged last
    tile), d = 3, K = 2, two draws + prediction at 130 sites (two extra tile rows), against LAPACK."""
    import scipy.linalg as sla
    n, d, K, m = 8100, 3, 2, 130
    X, y = synthetic_design(n, d, seed=99)
    Xt = np.random.default_rng(5).random((m, d))
    P = np.array([[0.6, 0.4, 2.0, 3.0, 4.0, 150.0, 180.0, 200.0],
                  [0.3, 0.7, 1.0, 1.5, 2.5, 120.0, 140.0, 260.0]])
    ll, beta, st = handle.loglik_batch(X, y, K, P, 1.3, 0, 0.0)
    mean, var, beta_p, st_p = handle.predict_batch(X, y, K, P[:1], Xt, 1.3)
    assert not st.any() and not st_p.any()
    w, Th = orc.unpack_params(P[0], K, d)
    R = orc.mixed_corr_matrix_general(X, w, Th)
    L = sla.cholesky(R, lower=True)
    zy = sla.solve_triangular(L, y, lower=True)
    z1 = sla.solve_triangular(L, np.ones(n), lower=True)
    b0 = (z1 @ zy) / (z1 @ z1)
    c = 1.3 * np.sum(w ** 2)
    want = -0.5 * (n * math.log(2 * math.pi) + n * math.log(c) + 2 * np.log(np.diag(L)).sum()
                   + np.sum((zy - b0 * z1) ** 2) / c)
    assert ll[0] == pytest.approx(want, rel=1e-9)
    assert beta[0] == pytest.approx(b0, rel=1e-7, abs=1e-10) and beta_p[0] == beta[0]
    # predict.post (HX:667-670) in the factor's metric for a few sites
    r = sum(w[c] ** 2 * np.exp(-(((Xt[:5, None, :] - X[None, :, :]) ** 2) * Th[c]).sum(axis=2)) for c in range(K)) / np.sum(w ** 2)
    Wt = sla.solve_triangular(L, r.T, lower=True)          # n x 5:  L^-1 r(x_t)
    zr = zy - b0 * z1
    np.testing.assert_allclose(mean[0, :5], b0 + zr @ Wt, rtol=1e-8, atol=1e-10)
    want_var = 1.3 * (1.0 - np.sum(Wt * Wt, axis=0) + (1.0 - z1 @ Wt) ** 2 / (z1 @ z1))
    np.testing.assert_allclose(var[0, :5], want_var, rtol=1e-6, atol=1e-9)


# ------------------------------------------------------------------------------- failure detection
def test_non_positive_definite_maps_to_na(handle):
    """HX:454-455: a singular R gives NA, not an abort.  A duplicated design point makes R
    exactly singular; status carries the 1-based pivot and the value is NaN."""
    D, y, _, _ = load_qian()
    Dd = D.copy()
    Dd[40] = Dd[3]
    P = np.stack([orc.params_from_iso(0.8, 0.3, 15.0, 4), orc.params_from_iso(0.7, 0.5, 25.0, 4)])
    ll, beta, st = handle.loglik_batch(Dd, y, 2, P, 10.0, 0, 0.0)
    assert np.all(st > 0) and np.all(st <= 64) and np.all(np.isnan(ll)) and np.all(np.isnan(beta))
    ok, _, st_ok = handle.loglik_batch(D, y, 2, P, 10.0, 0, 0.0)
    assert not st_ok.any() and np.all(np.isfinite(ok))
    from ccgp_amd.rsurface import CombinedGP
    r = CombinedGP("GV", handle=handle).logpost(Dd, [0.0, 2.0, 1.0], y, 10.0)
    assert math.isnan(r["val"]) and r["R_Inv"] is None
    # blocked path
    X, yy = synthetic_design(300, 3, seed=9)
    X[250] = X[17]
    Pb = orc.params_from_iso(0.6, 2.0, 9.0, 3)[None]
    llb, _, stb = handle.loglik_batch(X, yy, 2, Pb, 1.0, 0, 0.0)
    assert stb[0] > 0 and math.isnan(llb[0])


def test_chunked_batches_equal_one_pass(handle):
    """A batch larger than the device workspace is processed in chunks (ccgp_set_workspace_limit): same
    values, same per-draw status, ragged last chunk, a singular draw in the middle of a chunk."""
    X, y = synthetic_design(300, 3, seed=21)
    Xt = np.random.default_rng(2).random((40, 3))
    rng = np.random.default_rng(8)
    P = np.stack([orc.params_from_iso(rng.uniform(0.3, 0.9), rng.uniform(1, 3), rng.uniform(8, 20), 3) for _ in range(7)])
    Xs = X.copy()
    ll0, b0, st0 = handle.loglik_batch(Xs, y, 2, P, 1.0, 0, 0.0)
    m0, v0, _, _ = handle.predict_batch(Xs, y, 2, P, Xt, 1.0)
    try:
        handle.set_workspace_limit(4 << 20)                 # ~2 matrices of 384 x 512 doubles per pass
        ll1, b1, st1 = handle.loglik_batch(Xs, y, 2, P, 1.0, 0, 0.0)
        m1, v1, _, _ = handle.predict_batch(Xs, y, 2, P, Xt, 1.0)
        Xd = X.copy()
        Xd[120] = Xd[7]                                     # exactly singular: pivot 121 is 0 up to rounding
        lld, _, std = handle.loglik_batch(Xd, y, 2, P, 1.0, 0, 0.0)
    finally:
        handle.set_workspace_limit(200 << 30)
    np.testing.assert_array_equal(ll0, ll1)
    np.testing.assert_array_equal(b0, b1)
    np.testing.assert_array_equal(st0, st1)
    np.testing.assert_array_equal(m0, m1)
    np.testing.assert_array_equal(v0, v1)
    # the pivot of the duplicated row is 0 in exact arithmetic: it comes out non-positive (status = its
    # 1-based index, NaN value -- the reference's NA) or, for some draws, as a positive rounding residue
    assert set(std.tolist()) <= {0, 121} and np.count_nonzero(std) >= 4
    assert np.all(np.isnan(lld[std > 0]))


def test_edge_shapes(handle):
    D, y, Dt, _ = load_qian()
    ll, beta, st = handle.loglik_batch(D, y, 2, np.empty((0, 10)), 1.0)
    assert ll.shape == (0,)
    # n = 1 and n = 2
    for n in (1, 2, 3):
        P = orc.params_from_iso(0.8, 0.3, 15.0, 4)[None]
        w, Th = orc.unpack_params(P[0], 2, 4)
        got, gb, _ = handle.loglik_batch(D[:n], y[:n], 2, P, 2.0, 1, 9.0)
        assert got[0] == pytest.approx(orc.loglik_general(D[:n], y[:n], w, Th, 2.0, 1, 9.0)[0], rel=1e-12)
    # single test point, single draw
    from ccgp_amd.rsurface import CombinedGP
    t = CombinedGP("HX", handle=handle).prediction_table(Dt[:1], [(0.8, 0.3, 15.0)], D, 10.0, y)
    m, v = orc.predict_post_iso(Dt[0], D, y, 0.8, 0.3, 15.0, 10.0)
    assert t["mean"][0, 0] == pytest.approx(m, rel=1e-9) and t["var"][0, 0] == pytest.approx(v, rel=1e-8)
    # argument errors come back as codes, not crashes
    from ccgp_amd import api
    with pytest.raises(api.CcgpError):
        handle.loglik_batch(D, y, 9, np.ones((1, 45)), 1.0)


# ------------------------------------------------------------------------------- gradient extension
@pytest.mark.parametrize("case", ["maximin14", "qian", "gv90"])
def test_gradient_matches_finite_differences(handle, case):
    if case == "maximin14":
        X = load_maximin(14)
        y = np.array([orc.test_function_2d(a, b, 3) for a, b in X])
        rows = np.stack([orc.params_from_aniso(0.7, 2.0, 3.0, 4.0), np.array([0.6, 0.4, 3.0, 5.0, 20.0, 30.0])])
        K, s2, tol = 2, 0.4, 2e-4
    elif case == "qian":
        X, y, _, _ = load_qian()
        rows = np.stack([orc.params_from_iso(0.8, 0.3, 15.0, 4), orc.params_from_iso(0.7, 0.5, 25.0, 4)])
        K, s2, tol = 2, 10.0, 1e-5
    else:
        X, y, _, _ = load_gv(90)
        rows = orc.params_from_iso(0.7, 0.3, 15.0, 9)[None]
        K, s2, tol = 2, float(np.var(y, ddof=1)), 1e-5
    d = X.shape[1]
    ll, beta, grad, st = handle.loglik_grad_batch(X, y, K, rows, s2)
    assert not st.any()
    for b in range(rows.shape[0]):
        w, Th = orc.unpack_params(rows[b], K, d)
        assert ll[b] == pytest.approx(orc.loglik_general(X, y, w, Th, s2)[0], rel=1e-9)
        fd = orc.loglik_grad_fd(X, y, rows[b], K, d, s2)
        np.testing.assert_allclose(grad[b], fd, rtol=tol, atol=tol * np.abs(fd).max())


def test_register_resident_gradient_wide_design_three_components_and_a_failed_draw(handle):
    """The n <= 128 gradient (round 3: `small_reg_kernel<16, NB, NB + 1, gradient>`) beyond the reference's shapes:
    d = 20 (two groups of 16 per-dimension accumulators), K = 3, n = 45 (not a multiple of 16), a batch in which one
    draw is exactly singular (all scales 0: R = 11')."""
    from ccgp_amd import api
    rng = np.random.default_rng(31)
    n, d, K = 45, 20, 3
    X = rng.random((n, d))
    y = np.sin(3 * X[:, 0]) + X[:, 1:4].sum(axis=1)
    rows = np.stack([np.concatenate([[0.5, 0.3, 0.2], rng.uniform(0.05, 0.3, d), rng.uniform(0.5, 1.5, d), rng.uniform(3.0, 6.0, d)])
                     for _ in range(4)])
    rows[2, K:] = 0.0
    ll, beta, grad, st = handle.loglik_grad_batch(X, y, K, rows, 2.0)
    assert list(st != 0) == [False, False, True, False] and np.isnan(grad[2]).all() and np.isnan(ll[2])
    for b in (0, 1, 3):
        w, Th = orc.unpack_params(rows[b], K, d)
        assert ll[b] == pytest.approx(orc.loglik_general(X, y, w, Th, 2.0)[0], rel=1e-9)
        fd = orc.loglik_grad_fd(X, y, rows[b], K, d, 2.0)
        np.testing.assert_allclose(grad[b], fd, rtol=2e-5, atol=2e-5 * np.abs(fd).max())


# ------------------------------------------------------------------------------- 8(f)-4 entropy criteria
def test_entropy_criteria_over_candidate_designs(handle):
    """Batch Sequential ME Design.R:856-877: -det(R.mixed) for candidate designs, and the augmented
    (Schur complement) criterion, against the oracle's literal restatement."""
    from ccgp_amd.rsurface import CombinedGP
    gp = CombinedGP("BSQ", handle=handle)
    rng = np.random.default_rng(12)
    p, t1, t2 = 0.7, 2.0, 16.0
    designs = -1.0 + 2.0 * rng.random((37, 14, 2))          # 37 candidate 14-point designs in [-1,1]^2
    got = gp.Entropy_batch(designs, p, t1, t2)
    want = np.array([orc.entropy(D, p, t1, t2) for D in designs])
    np.testing.assert_allclose(got, want, rtol=1e-9)
    assert gp.Entropy(designs[3], p, t1, t2) == pytest.approx(want[3], rel=1e-9)
    D_old, D_new = designs[0], -1.0 + 2.0 * rng.random((7, 2))
    # cross.corr.matrix (BSQ:835-848), the override a script that calls it directly gets: n.new x n.old
    C = gp.cross_corr_matrix(D_old, D_new, t2)
    assert C.shape == (7, 14)
    np.testing.assert_allclose(C, orc.cross_corr_matrix(D_old, D_new, t2), rtol=1e-12, atol=1e-300)
    got_aug = gp.Augmented_Mixed_Entropy(D_old, D_new, p, t1, t2)
    assert got_aug == pytest.approx(orc.augmented_mixed_entropy(D_old, D_new, p, t1, t2), rel=1e-8)
    # 21-point designs (14 + 7, the script's second stage) and a 9-D case on the G = 16 template
    d21 = -1.0 + 2.0 * rng.random((5, 21, 2))
    np.testing.assert_allclose(gp.Entropy_batch(d21, p, t1, t2), [orc.entropy(D, p, t1, t2) for D in d21], rtol=1e-9)
    d9 = rng.random((3, 90, 9))
    np.testing.assert_allclose(gp.Entropy_batch(d9, 0.7, 0.3, 15.0), [orc.entropy(D, 0.7, 0.3, 15.0) for D in d9],
                               rtol=1e-8)


def test_config1_matern_1d_surface_vs_oracle(handle):
    """BASELINE config 1 on the device: the 1-D script's Matern(nu = 5) surface (D1:348-389, D1:575-641,
    D1:794-812) through ccgp_set_kernel, against the oracle's besselK restatement and the d1 golden
    fixture.  n = 8, so the correlation matrix is well conditioned only for small scale parameters;
    tolerance 1e-9 on the log-posterior (cond(R) eps, as in R), 1e-12 on the correlations."""
    from ccgp_amd.rsurface import CombinedGP1D
    g = golden("d1_golden.json")
    X = np.array(g["X"])
    y = np.array(g["y"])
    nu = g["nu"]
    gp = CombinedGP1D(nu, handle=handle)
    R = gp.corr_matrix(nu, X.reshape(-1, 1), 0.7)
    np.testing.assert_allclose(R, orc.corr_matrix_matern(nu, X.reshape(-1, 1), 0.7), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(gp.corr_vec(0.37, X, 0.4, nu), orc.corr_vec_matern(0.37, X, 0.4, nu), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(gp.corr_matrix(2.5, X, 0.9), orc.corr_matrix_matern(2.5, X.reshape(-1, 1), 0.9),
                               rtol=1e-12, atol=1e-15)                      # another smoothness through the same object
    Rm = gp.Mixed_corr_matrix(X, 0.8, 0.3, 0.9)
    want = (0.64 * orc.corr_matrix_matern(nu, X.reshape(-1, 1), 0.3) + 0.04 * orc.corr_matrix_matern(nu, X.reshape(-1, 1), 0.9)) / 0.68
    np.testing.assert_allclose(Rm, want, rtol=1e-12, atol=1e-15)
    for c in g["cases"]:
        lp = gp.logpost(X, c["theta_t"], y, c["sigma2"], nu)
        assert lp["val"] == pytest.approx(c["val"], rel=1e-9)
        assert lp["beta"] == pytest.approx(c["beta"], rel=1e-8, abs=1e-10)
        digest_close(lp["R_Inv"], c["R_inv"], 1e-6)      # cond(R) ~ 1e7 for these draws: R.Inv to cond eps
    # predictions: (draw x test point) tables against D1:794-812 recomputed by the oracle
    draws = [(0.8, 0.25, 0.6), (0.6, 0.15, 0.9)]
    xt = np.array([0.11, 0.52, 0.93])
    s2 = 1.3
    t = gp.prediction_table(xt, draws, X, s2, y)
    for i, (p, t1, t2) in enumerate(draws):
        for j, x in enumerate(xt):
            m, v = orc.predict_post_1d(x, X.reshape(-1, 1), y, p, t1, t2, s2, nu)
            assert t["mean"][i, j] == pytest.approx(m, rel=1e-8, abs=1e-10)
            assert t["var"][i, j] == pytest.approx(v, rel=1e-7, abs=1e-10)
    # the frame row predict.post consumes (D1:794-812), literally
    frame = gp.factors_frame_from_draws(draws[:1], X, s2, y)
    mv = gp.predict_post(0.52, X, frame[0], s2, nu)
    m, v = orc.predict_post_1d(0.52, X.reshape(-1, 1), y, *draws[0], s2, nu)
    assert mv[0, 0] == pytest.approx(m, rel=1e-8) and mv[0, 1] == pytest.approx(v, rel=1e-7, abs=1e-10)
    # the Gaussian family is back for everybody else, and shapes the Matern family does not have fail loudly
    from ccgp_amd import api
    handle.set_kernel(api.KERNEL_MATERN, 5.0)
    with pytest.raises(api.CcgpError):
        handle.corr_matrix(np.random.default_rng(0).random((5, 2)), 1.0)
    with pytest.raises(api.CcgpError):
        handle.set_kernel(api.KERNEL_MATERN, 0.5)
    handle.set_kernel(api.KERNEL_GAUSS)
    Dq, _, _, _ = load_qian()
    np.testing.assert_allclose(handle.corr_matrix(Dq, 0.3), orc.corr_matrix_iso(Dq, 0.3), rtol=1e-12)


def test_two_family_1d_surface_vs_fixture(handle):
    """The two-family 1-D script (D1F) on the device: Matern(nu = 5) + cubic spline through
    CCGP_KERNEL_MATERN_SPLINE, against the d1f fixture (oracle restatement of D1F:346-601, 737-754),
    including the un-normalised corr.vec.combined of D1F:479 and the predictions built on it."""
    from ccgp_amd.rsurface import CombinedGP1DTwoFamilies
    g = golden("d1f_golden.json")
    X, y, nu = np.array(g["X"]), np.array(g["y"]), g["nu"]
    gp = CombinedGP1DTwoFamilies(nu, handle=handle)
    digest_close(gp.corr_matrix_spline(X, 0.45), g["R_spline"], 1e-13)
    np.testing.assert_allclose(gp.corr_vec_spline(0.41, X, 0.45), orc.corr_vec_spline(0.41, X, 0.45), rtol=1e-13, atol=1e-16)
    np.testing.assert_allclose(gp.corr_matrix_Matern(nu, X, 0.5), orc.corr_matrix_matern(nu, X.reshape(-1, 1), 0.5), rtol=1e-12)
    digest_close(gp.corr_matrix_combined(X, 0.7, 0.5, 0.6, nu), g["R_combined"], 1e-12)
    np.testing.assert_allclose(gp.corr_vec_combined(0.41, X, 0.7, 0.5, 0.6, nu), g["r_combined"], rtol=1e-12)
    xt = np.array(g["xt"])
    for c in g["cases"]:
        lp = gp.logpost(X, c["theta_t"], y, c["sigma2"], nu)
        assert lp["val"] == pytest.approx(c["val"], rel=1e-10)
        assert lp["beta"] == pytest.approx(c["beta"], rel=1e-9, abs=1e-12)
        digest_close(lp["R_Inv"], c["R_inv"], 1e-9)
        t = c["theta_t"]
        draw = (1.0 / (1.0 + math.exp(-t[2])), math.exp(t[0]), math.exp(t[1]))
        tab = gp.prediction_table(xt, [draw], X, c["sigma2"], y)
        np.testing.assert_allclose(tab["mean"][0], c["pred_mean"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(tab["var"][0], c["pred_var"], rtol=1e-9, atol=1e-12)
    from ccgp_amd import api
    handle.set_kernel(api.KERNEL_MATERN_SPLINE, 5.0)
    with pytest.raises(api.CcgpError):                       # the pair is K = 2 by definition
        handle.mixed_corr_matrix(X.reshape(-1, 1), 3, np.array([0.3, 0.3, 0.4, 0.5, 0.6, 0.7]))
    handle.set_kernel(api.KERNEL_GAUSS)


def test_adv_prediction_tables_batched_and_as_written(handle):
    """ADV trains with R2 = corr.matrix.ISO(D, lambda) (ADV:417) and predicts with theta1 * (1 + lambda)
    (ADV:672).  The batched table uses the training kernel for r; as_written=True reproduces the script."""
    from ccgp_amd.rsurface import CombinedGP
    gd = golden("adv_golden.json")
    D14, y14 = load_maximin(14), np.array(gd["y"])
    Dt = np.array([[0.21, 0.4], [0.66, 0.12], [0.5, 0.93]])
    draws = np.array([[0.8, 0.9, 6.0], [0.65, 1.4, 9.0]])                  # (p, theta1, lambda)
    s2 = gd["sigma2"]
    gp = CombinedGP("ADV", handle=handle)
    batched = gp.prediction_table(Dt, draws, D14, s2, y14)
    literal = gp.prediction_table(Dt, draws, D14, s2, y14, as_written=True)
    for s, (p, t1, lam) in enumerate(draws):
        R_inv = orc.solve_inverse(orc.mixed_corr_matrix_iso(D14, p, t1, lam))
        beta = orc.beta_mle(R_inv, y14)
        mf, v1, v2 = orc.factors(R_inv, beta, y14)
        for t in range(Dt.shape[0]):
            consistent = orc.predict_post_from_factors(orc.mixed_corr_vec_iso(Dt[t], D14, p, t1, lam), beta, mf, v1, v2, R_inv, s2)
            written = orc.predict_post_from_factors(orc.mixed_corr_vec_iso(Dt[t], D14, p, t1, t1 * (1.0 + lam)), beta, mf, v1, v2,
                                                    R_inv, s2)
            assert batched["mean"][s, t] == pytest.approx(consistent[0], rel=1e-7)
            assert batched["var"][s, t] == pytest.approx(consistent[1], rel=1e-5, abs=1e-9)
            assert literal["mean"][s, t] == pytest.approx(written[0], rel=1e-7)
            assert literal["var"][s, t] == pytest.approx(written[1], rel=1e-5, abs=1e-9)
    assert np.max(np.abs(batched["mean"] - literal["mean"])) > 1e-3     # the two really are different predictors


# ------------------------------------------------------------------------------- round-2 coverage gaps
def test_every_ground_vibrations_pair(handle):
    """BASELINE config 5 runs ALL train/test pairs the reference ships (9 of size 50, 8 of size 90); the
    sample-1 fixtures above cover every test site, this one covers every training set."""
    from ccgp_amd.rsurface import CombinedGP
    g = golden("gv_all_golden.json")
    gp = CombinedGP("GV", handle=handle)
    assert len(g["sets"]) == 17
    for s in g["sets"]:
        Dg, yg, Dtg, _ = load_gv(s["size"], s["sample"])
        t = gp.prediction_table(Dtg[::g["site_step"]], g["draws"], Dg, s["sigma2"], yg)
        assert not t["status"].any()
        np.testing.assert_allclose(t["mean"], s["mean"], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(t["var"], s["var"], rtol=1e-8, atol=1e-10 * s["sigma2"])
        np.testing.assert_allclose(t["beta"], s["beta"], rtol=1e-8, atol=1e-11)


def test_multi_chunk_batch_at_n4096_equals_one_pass(handle):
    """The blocked path's chunking at the benchmark's size: 20 draws at n = 4096 in one pass and in chunks of 7, 7, 6
    (workspace capped at 1 GiB): identical values, betas and status."""
    n, d, K = 4096, 5, 3
    X, y = synthetic_design(n, d, seed=20140101)
    rng = np.random.default_rng(17)
    P = np.empty((20, K + K * d))
    for b in range(20):
        th = np.exp(rng.uniform(np.log(0.5), np.log(50.0), size=(K, d)))
        th[-1] = np.maximum(th[-1], 20.0)
        P[b] = np.concatenate([0.15 + 0.55 * rng.dirichlet(np.ones(K)), th.ravel()])
    ll0, b0, st0 = handle.loglik_batch(X, y, K, P, 1.0)
    try:
        handle.set_workspace_limit(1 << 30)
        ll1, b1, st1 = handle.loglik_batch(X, y, K, P, 1.0)
    finally:
        handle.set_workspace_limit(200 << 30)
    assert not st0.any() and np.all(np.isfinite(ll0))
    np.testing.assert_array_equal(ll0, ll1)
    np.testing.assert_array_equal(b0, b1)
    np.testing.assert_array_equal(st0, st1)
    # an evaluation does not depend on the batch it travels in (8 and 20 matrices take different launch shapes:
    # whole tiles vs tail strips), which is what makes results identical at 1, 2, 4, 8 ranks
    ll2, b2, _ = handle.loglik_batch(X, y, K, P[:8], 1.0)
    np.testing.assert_array_equal(ll2, ll0[:8])
    np.testing.assert_array_equal(b2, b0[:8])


def test_config3_grid_as_bundled(handle):
    """BASELINE config 3 on `2D Codes and Designs/hyperpars.matrix.txt` AS BUNDLED (the x16 fixture above is the
    well-conditioned variant): the scales were tuned for 14 points, on maximin-100 cond(Sigma) runs from 2e9 to
    8e17.  Three grid rows x 1728 nodes against the oracle: agreement to cond * eps wherever the problem is
    solvable in fp64 at all; an evaluation is NaN exactly when its status is non-zero; and the device and the
    reference arithmetic (mnormt::dmnorm -> chol, ADV:573, which ABORTS the R script there) disagree about
    success only where cond > 1e15, i.e. where the outcome of a Cholesky is decided by rounding."""
    g = golden("cfg3_bundled_golden.json")
    D, y = load_maximin(100), np.array(g["y"])
    H = load_hyper("adv")
    eps = np.finfo(float).eps
    for r in g["rows"]:
        vals, arg, logs = handle.grid_marginal(D, y, g["sigma2"], H[r["row"]:r["row"] + 1], g["N"], g["tau"], False,
                                               aniso_lambda=g["aniso_lambda"], want_logs=True)
        got = logs[0]
        cond = np.array(r["cond"])
        want = np.array([np.nan if v is None else v for v in r["values"]])
        ok_dev, ok_ref = np.isfinite(got), np.isfinite(want)
        both = ok_dev & ok_ref
        tol = np.maximum(1e-9, 200.0 * cond * eps)
        solvable = both & (cond < 1e14)
        assert solvable.sum() > 0.5 * got.size
        assert np.all(np.abs(got[solvable] - want[solvable]) <= tol[solvable] * np.abs(want[solvable]))
        assert np.all(cond[ok_dev != ok_ref] > 1e15), cond[ok_dev != ok_ref]
        assert np.all(cond[~ok_dev] > 1e15)
        # the row's mean of exp() is NaN as soon as one node failed (R's mean() propagates NA; here the script would
        # have stopped inside dmnorm), finite otherwise
        assert np.isnan(vals[0]) == (not ok_dev.all())
    # per-evaluation status through the batch entry point: NaN <=> status != 0
    u = orc.runif_halton(g["N"])
    row0 = H[g["rows"][0]["row"]]
    th1, th2 = orc.qigamma(u, row0[0], row0[1]), orc.qigamma(u, row0[2], row0[3])
    lam = g["aniso_lambda"]
    P = np.column_stack([u, 1 - u, th1, th2, (1 + lam) * th1, (1 + lam) * th2])
    ll, _, st = handle.loglik_batch(D, y, 2, P, g["sigma2"], 1, g["tau"] ** 2)
    np.testing.assert_array_equal(np.isnan(ll), st != 0)


def test_widest_design_and_largest_chunks(handle):
    """Limits the header admits: d = 64 inputs (the covariance kernel's LDS staging passes the 64 KiB a kernel
    gets without asking: round-1 advisor finding), and a blocked-path batch of more draws than a grid dimension
    holds (the chunk index is a grid y / z coordinate: chunks are capped at 65535)."""
    from ccgp_amd import api
    rng = np.random.default_rng(3)
    n, d, K = 70, 64, 2
    X = rng.random((n, d))
    y = np.sin(X.sum(axis=1))
    row = np.concatenate([[0.7, 0.3], rng.uniform(0.01, 0.05, d), rng.uniform(0.3, 0.6, d)])
    w, Th = orc.unpack_params(row, K, d)
    R = handle.mixed_corr_matrix(X, K, row)
    np.testing.assert_allclose(R, orc.mixed_corr_matrix_general(X, w, Th), rtol=1e-12, atol=1e-15)
    r = handle.mixed_corr_cross(X[:3] + 0.01, X, K, row)
    want = np.array([orc.mixed_corr_vec_general(x, X, w, Th) for x in X[:3] + 0.01])
    np.testing.assert_allclose(r, want, rtol=1e-12, atol=1e-15)
    ll, beta, st = handle.loglik_batch(X, y, K, row[None], 1.0)          # d = 64 does not fit the fused evaluators' LDS budget
    wl, wb = orc.loglik_general(X, y, w, Th, 1.0)
    assert st[0] == 0 and ll[0] == pytest.approx(wl, rel=1e-9) and beta[0] == pytest.approx(wb, rel=1e-8)
    # 70 000 draws of the 1-D Matern family (always the materialised-matrix path): two chunks
    Xs = np.linspace(0.05, 0.95, 8)[:, None]
    ys = np.sin(10.0 * Xs[:, 0])
    B = 70000
    P = np.column_stack([rng.uniform(0.3, 0.9, B), np.zeros(B), rng.uniform(0.2, 0.6, B), rng.uniform(0.05, 0.15, B)])
    P[:, 1] = 1.0 - P[:, 0]
    try:
        handle.set_kernel(api.KERNEL_MATERN, 5.0)
        ll, beta, st = handle.loglik_batch(Xs, ys, 2, P, 1.0)
        pick = np.array([0, 1, 65534, 65535, 65536, B - 1])
        ll2, beta2, st2 = handle.loglik_batch(Xs, ys, 2, P[pick], 1.0)
    finally:
        handle.set_kernel(api.KERNEL_GAUSS, 0.0)
    assert not st.any() and np.all(np.isfinite(ll))
    np.testing.assert_array_equal(ll[pick], ll2)
    np.testing.assert_array_equal(beta[pick], beta2)
    with pytest.raises(api.CcgpError):
        handle.set_option(99, 1)
    for removed in (0, 1, 6):       # CCGP_OPT_UPDATE_STRIPS / SMALL_LDS / FUSED_COV of rounds 1 - 4
        with pytest.raises(api.CcgpError):
            handle.set_option(removed, 1)


@pytest.mark.parametrize("n", [65, 72, 81, 96, 100, 104])
def test_one_wave_per_matrix_and_the_16x16_grid_give_the_same_bits(handle, n):
    """64 < n <= 104 runs one wave per matrix on the 8 x 8 thread grid (up to 13 x 13 blocks per thread) since round 4;
    OPT_SMALL_GRID16 selects the 16 x 16 grid of rounds 1 - 3.  Every matrix entry sees the same operations in the same
    order on both, so log-likelihood, beta and status must agree bit for bit -- in both mean modes, including a draw whose
    factorisation fails -- and both must agree with the oracle."""
    from ccgp_amd import api
    d, K = 2, 2
    X, y = synthetic_design(n, d, seed=7 * n)
    rng = np.random.default_rng(n)
    B = 70                                          # more than 64 draws: the throughput dispatch, not the latency one
    P = np.column_stack([rng.uniform(0.3, 0.9, B), rng.uniform(0.1, 0.7, B), rng.uniform(0.5, 3.0, (B, d)),
                         rng.uniform(20.0, 60.0, (B, d))])
    P[5, 2:] = 0.0                                  # R = 11': singular
    for mode, tau2 in ((api.MEAN_PROFILE_BETA, 0.0), (api.MEAN_ZERO_PLUS_TAU2, 9.0)):
        a = handle.loglik_batch(X, y, K, P, 1.3, mode, tau2)
        handle.set_option(api.OPT_SMALL_GRID16, 1)
        try:
            b = handle.loglik_batch(X, y, K, P, 1.3, mode, tau2)
        finally:
            handle.set_option(api.OPT_SMALL_GRID16, 0)
        for u, v in zip(a, b):
            np.testing.assert_array_equal(u, v)
        assert a[2][5] != 0 and np.isnan(a[0][5]) and (np.delete(a[2], 5) == 0).all()
        w, Th = orc.unpack_params(P[0], K, d)
        want = orc.loglik_general(X, y, w, Th, 1.3, mode, tau2)[0]
        assert a[0][0] == pytest.approx(want, rel=1e-9)


def test_large_tables_come_back_in_slices_unchanged(handle):
    """Host-pointer ccgp_predict_batch with 2 x 840 KB of tables: the results cross PCIe in four slices, each copied out of
    the pinned buffer while the next is on its way (capi.hip: pull).  Every piece (mean, var, beta, status) straddles slice
    boundaries somewhere.  The same draws in chunks of 100 (240 KB per call: one slice) must give the same bits -- every
    draw is computed by its own workgroup, so only the way back differs."""
    D, y, Dt, _ = load_gv(50)
    rng = np.random.default_rng(17)
    S = 700
    P = np.array([np.concatenate([[p, 1 - p], np.full(9, a), np.full(9, b)]) for p, a, b in
                  zip(rng.uniform(0.5, 0.9, S), rng.uniform(0.2, 0.5, S), rng.uniform(10, 20, S))])
    P[123, 2:] = 0.0                                   # one failing draw: NaN row, status != 0
    mean, var, beta, st = handle.predict_batch(D, y, 2, P, Dt, 10.0)
    for lo in range(0, S, 100):
        m2, v2, b2, s2 = handle.predict_batch(D, y, 2, P[lo:lo + 100], Dt, 10.0)
        np.testing.assert_array_equal(mean[lo:lo + 100], m2)
        np.testing.assert_array_equal(var[lo:lo + 100], v2)
        np.testing.assert_array_equal(beta[lo:lo + 100], b2)
        np.testing.assert_array_equal(st[lo:lo + 100], s2)
    assert st[123] != 0 and np.isnan(mean[123]).all() and np.isfinite(np.delete(mean, 123, axis=0)).all()


def test_one_wave_per_matrix_random_shapes_against_the_oracle(handle):
    """Random (n, d, K) in the range the one-wave 8 x 8 grid serves (64 < n <= 104), including K = 3 and wide designs
    (d = 20: still four matrices per workgroup in LDS; d = 40: falls back to the 16 x 16 grid): the 16 x 16 grid gives
    the same bits, the oracle the same numbers."""
    from ccgp_amd import api
    rng = np.random.default_rng(2024)
    for trial in range(10):
        n = int(rng.integers(65, 105))
        d = int(rng.choice([1, 2, 3, 5, 9, 20, 40]))
        K = int(rng.choice([1, 2, 3]))
        X, y = synthetic_design(n, d, seed=100 + trial)
        B = 66
        W = rng.uniform(0.3, 0.9, (B, K))
        rough = 2.0 * n ** (2.0 / d) / d                      # theta h^2 ~ 1 at the typical spacing h = n^(-1/d)
        Th = np.exp(rng.uniform(np.log(0.01 * rough), np.log(0.3 * rough), (B, K * d)))
        Th[:, -d:] = rng.uniform(rough, 2.0 * rough, (B, d))   # the roughest component keeps R positive definite
        P = np.column_stack([W, Th])
        mode, tau2 = ((api.MEAN_PROFILE_BETA, 0.0), (api.MEAN_ZERO_PLUS_TAU2, 4.0))[trial % 2]
        a = handle.loglik_batch(X, y, K, P, 0.9, mode, tau2)
        handle.set_option(api.OPT_SMALL_GRID16, 1)
        try:
            b = handle.loglik_batch(X, y, K, P, 0.9, mode, tau2)
        finally:
            handle.set_option(api.OPT_SMALL_GRID16, 0)
        for u, v in zip(a, b):
            np.testing.assert_array_equal(u, v, err_msg="n=%d d=%d K=%d" % (n, d, K))
        assert (a[2] == 0).sum() >= B - 3, (n, d, K, a[2])     # a numerically singular draw may occur; it must be rare
        for bidx in [i for i in (0, B - 1) if a[2][i] == 0]:
            w, T = orc.unpack_params(P[bidx], K, d)
            want_ll, want_beta = orc.loglik_general(X, y, w, T, 0.9, mode, tau2)
            cond = np.linalg.cond(orc.mixed_corr_matrix_general(X, w, T))
            assert a[0][bidx] == pytest.approx(want_ll, rel=max(1e-10, 20 * cond * np.finfo(float).eps)), (n, d, K, cond)
            if mode == api.MEAN_PROFILE_BETA:
                assert a[1][bidx] == pytest.approx(want_beta, rel=max(1e-9, 50 * cond * np.finfo(float).eps), abs=1e-10)


@pytest.mark.parametrize("n,d,K", [(30, 5, 3), (50, 8, 4), (100, 3, 7), (20, 1, 2), (77, 9, 2)])
def test_small_n_gradient_pass_structure(handle, n, d, K):
    """The round-4 gradient contraction carries QG x KG accumulators per pass over the pairs: 3 components x 8 dimensions
    when d <= 8 (K = 3, d = 5: one pass; K = 4, d = 8 and K = 7, d = 3: several component groups), 1 x 16 otherwise
    (d = 9).  Every case against central differences of the oracle's log-likelihood."""
    X, y = synthetic_design(n, d, seed=11 * n + d)
    rng = np.random.default_rng(n + K)
    rough = 2.0 * n ** (2.0 / d) / d
    W = rng.uniform(0.3, 0.9, K)
    Th = np.exp(rng.uniform(np.log(0.02 * rough), np.log(0.3 * rough), (K, d)))
    Th[-1] = rng.uniform(rough, 2.0 * rough, d)
    row = np.concatenate([W, Th.ravel()])
    ll, beta, grad, st = handle.loglik_grad_batch(X, y, K, np.stack([row, row * 1.01]), 0.8)
    assert not st.any() and np.isfinite(grad).all()
    w, T = orc.unpack_params(row, K, d)
    assert ll[0] == pytest.approx(orc.loglik_general(X, y, w, T, 0.8)[0], rel=1e-9)
    fd = orc.loglik_grad_fd(X, y, row, K, d, 0.8)
    np.testing.assert_allclose(grad[0], fd, rtol=5e-5, atol=5e-5 * np.abs(fd).max())


# ------------------------------------------------------------------------------- a10 / a11: kept-factor prediction (round 5)
@pytest.mark.parametrize("n,d,K,S,m", [(8, 1, 2, 5, 3), (20, 3, 1, 30, 70), (50, 9, 2, 120, 150), (64, 4, 2, 100, 14), (65, 2, 3, 40, 129),
                                       (90, 9, 2, 150, 110), (100, 2, 2, 64, 200), (104, 5, 3, 33, 65)])
def test_kept_factor_prediction_has_the_bits_of_the_extra_row_scheme(handle, n, d, K, S, m):
    """predict.post tables (HX:655-673 over HX:686-725) at n <= 104: since round 5 each draw is factorised ONCE (the likelihood
    kernel keeps L', 1/d, z'_y, z'_1 in HBM), the test sites' correlation vectors are formed beside it on a second stream
    (lane = site) and a third kernel does the forward substitutions (lane = site, L' by broadcast LDS reads).  Rounds 2 - 4
    carried the sites through the elimination as extra rows, 30 / 62 per factorisation.  Every (draw, site) entry sees the same
    operations in the same order under both, so mean, variance, beta and status agree bit for bit -- including a draw whose
    matrix cannot be factorised -- and both agree with the oracle."""
    from ccgp_amd import api
    rng = np.random.default_rng(1000 * n + m)
    X = rng.uniform(size=(n, d))
    y = np.sin(2 * np.pi * X).sum(axis=1)
    P = np.empty((S, K + K * d))
    for b in range(S):
        w = 0.15 + 0.55 * rng.dirichlet(np.ones(K))
        th = np.exp(rng.uniform(np.log(0.5), np.log(30.0), size=(K, d)))
        th[K - 1] = np.maximum(th[K - 1], 25.0)
        P[b] = np.concatenate([w, th.ravel()])
    P[S // 2, K:] = 1e-5
    Xt = rng.uniform(size=(m, d))
    res = {}
    for opt in (0, 1):
        handle.set_option(api.OPT_PREDICT_FACTOR, opt)
        try:
            res[opt] = handle.predict_batch(X, y, K, P, Xt, 1.7)
        finally:
            handle.set_option(api.OPT_PREDICT_FACTOR, 1)
    for a, b in zip(res[0], res[1]):
        np.testing.assert_array_equal(np.asarray(a), np.asarray(b))
    st = np.asarray(res[1][3])
    assert (st != 0).sum() <= 2 and np.isnan(res[1][0][st != 0]).all() and np.isfinite(res[1][0][st == 0]).all()
    if d < 9:
        assert st[S // 2] != 0          # the nearly constant correlation matrix is singular to working precision
    b = 0 if S // 2 else 1
    w, Th = orc.unpack_params(P[b], K, d)
    R_inv = orc.solve_inverse(orc.mixed_corr_matrix_general(X, w, Th))
    beta = orc.beta_mle(R_inv, y)
    mf, v1, v2 = orc.factors(R_inv, beta, y)
    for t in (0, m - 1):
        mean, var = orc.predict_post_from_factors(orc.mixed_corr_vec_general(Xt[t], X, w, Th), beta, mf, v1, v2, R_inv, 1.7)
        assert res[1][0][b, t] == pytest.approx(mean, rel=1e-7, abs=1e-8)
        assert res[1][1][b, t] == pytest.approx(var, rel=1e-5, abs=1e-8)


def test_kept_factor_prediction_in_chunks_of_draws(handle):
    """The factor blocks and correlation vectors live in the handle's workspace; when the limit does not hold them for every
    draw the call runs in chunks of draws -- same bits."""
    from ccgp_amd import api
    rng = np.random.default_rng(5)
    n, d, K, S, m = 50, 4, 2, 300, 150
    X = rng.uniform(size=(n, d))
    y = np.cos(3 * X).sum(axis=1)
    P = np.column_stack([rng.uniform(0.3, 0.9, S), np.zeros(S), np.exp(rng.uniform(-1, 2, size=(S, d))), np.full((S, d), 30.0)])
    P[:, 1] = 1 - P[:, 0]
    Xt = rng.uniform(size=(m, d))
    whole = handle.predict_batch(X, y, K, P, Xt, 1.0)
    handle.set_workspace_limit(20 << 20)        # ~ 100 KB per draw: chunks of about a hundred
    try:
        chunked = handle.predict_batch(X, y, K, P, Xt, 1.0)
    finally:
        handle.set_workspace_limit(200 << 30)
    for a, b in zip(whole, chunked):
        np.testing.assert_array_equal(np.asarray(a), np.asarray(b))
