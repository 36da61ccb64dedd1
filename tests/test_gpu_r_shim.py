"""r/ccgp_shim.c EXECUTED: every `.Call` routine of the R shim, dispatched through its own registration table, on the
real libccgp.so, against a functional mock of the R C API (tests/r_mock/: there is no R in the container or on the
GPU boxes).  One test per routine listed in INTEGRATION.md section 3.  What is checked for each: the results are the
ones `api.Handle` (ctypes -> the same C ABI) returns, shapes / `dim` / list layout are what r/ccgp.R indexes, a NaN from
the device becomes R's NA_real_, a negative return code becomes a warning plus NA (never an Rf_error out of a device
call), and the mock's bookkeeping stays clean (PROTECT balance, nothing left unprotected across an allocation, no
REAL() on a non-double).  Reference return shapes: `logpost` -> list(val, beta, R.Inv) (Heat Exchanger Emulator/Combined GP
Heat Exchanger.R:441-466), `choose.hyperpars` -> list(pars, likelihoods) (:584-595), `predict.post` -> cbind(mean, var)
(:655-673)."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, load_gv, load_hyper, load_maximin, load_qian, synthetic_design

sys.path.insert(0, os.path.join(ROOT, "tests", "r_mock"))
import rmock  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    os.environ.pop("CCGP_DEVICES", None)
    r = rmock.MockR()
    yield r
    r.unload()


@pytest.fixture(autouse=True)
def clean(R):
    R.reset()
    yield
    R.assert_clean()


def iso_row(p, t1, t2, d):
    return np.concatenate([[p, 1 - p], np.full(d, t1), np.full(d, t2)])


def test_corr_matrix_and_cross(R, handle):
    D, y, Dt, _ = load_qian()
    th = np.array([0.3, 0.5, 0.7, 1.1])
    got = R.dot_call("ccgp_R_corr_matrix", R.real(D), R.real(th))
    assert got.shape == (64, 64) and np.array_equal(got, handle.corr_matrix(D, th))
    got = R.dot_call("ccgp_R_corr_cross", R.real(Dt), R.real(D), R.real(th))
    assert got.shape == (14, 64) and np.array_equal(got, handle.corr_cross(Dt, D, th))
    # corr.vec: x.new as a 1 x d matrix (r/ccgp.R builds it with matrix(as.double(x), nrow = 1))
    one = R.dot_call("ccgp_R_corr_cross", R.real(Dt[:1]), R.real(D), R.real(th))
    assert one.shape == (1, 64) and np.array_equal(one[0], got[0])
    assert R.warnings() == []


def test_mixed_corr_matrix_and_cross(R, handle):
    D, y, Dt, _ = load_qian()
    row = iso_row(0.8, 0.3, 15.0, 4)
    got = R.dot_call("ccgp_R_mixed_corr_matrix", R.real(D), R.integer(2), R.real(row))
    assert np.array_equal(got, handle.mixed_corr_matrix(D, 2, row))
    got = R.dot_call("ccgp_R_mixed_corr_cross", R.real(Dt), R.real(D), R.integer(2), R.real(row))
    assert got.shape == (14, 64) and np.array_equal(got, handle.mixed_corr_cross(Dt, D, 2, row))


def test_negative_return_code_is_a_warning_and_na_not_an_error(R):
    D, _, _, _ = load_qian()
    got = R.dot_call("ccgp_R_mixed_corr_matrix", R.real(D), R.integer(9), R.real(np.ones(9 + 36)))   # K = 9 > 8
    assert got.shape == (64, 64) and R.is_na(got).all()
    w = R.warnings()
    assert len(w) == 1 and "libccgp error -1" in w[0] and "bad argument" in w[0]


def test_logpost_list_val_beta_rinv(R, handle):
    from ccgp_amd import api
    D, y, _, _ = load_qian()
    s2 = float(np.var(y, ddof=1))
    theta_t = np.array([np.log(0.3), np.log(15.0), np.log(4.0)])
    pars = np.array([7.0, 3.0, 3.0, 28.0])
    got = R.dot_call("ccgp_R_logpost", R.real(D), R.real(theta_t), R.real(y), R.real(s2), R.integer(api.PRIOR_INVGAMMA),
                     R.real(pars), R.integer(1))
    assert list(got) == ["val", "beta", "R.Inv", "loglik"]            # r/ccgp.R reads r$val, r$beta, r$R.Inv, r$loglik
    want = handle.logpost(D, y, s2, api.PRIOR_INVGAMMA, theta_t, pars)
    assert got["val"][0] == want["val"] and got["beta"][0] == want["beta"] and got["loglik"][0] == want["loglik"]
    assert got["R.Inv"].shape == (64, 64) and np.array_equal(got["R.Inv"], want["R_inv"])
    # scripts whose logpost takes no prior parameters pass NULL (GV:429-454)
    got = R.dot_call("ccgp_R_logpost", R.real(D), R.real(theta_t), R.real(y), R.real(s2), R.integer(api.PRIOR_GV), R.null(),
                     R.integer(1))
    assert got["val"][0] == handle.logpost(D, y, s2, api.PRIOR_GV, theta_t)["val"]
    # slim frame (r/ccgp.R, ccgp.slim.frame): solve(R) is neither formed nor shipped, Metro stores the 1 x 1 placeholder
    slim = R.dot_call("ccgp_R_logpost", R.real(D), R.real(theta_t), R.real(y), R.real(s2), R.integer(api.PRIOR_GV), R.null(),
                      R.integer(0))
    assert slim["R.Inv"].shape == (1,) and slim["R.Inv"][0] == 0.0
    assert abs(slim["val"][0] - got["val"][0]) <= 1e-12 * abs(got["val"][0]) and abs(slim["beta"][0] - got["beta"][0]) <= 1e-12
    assert R.warnings() == []


def test_logpost_singular_design_gives_na_like_try_solve(R):
    """try(solve(R), silent = TRUE) -> R.Inv <- NA (HX:454-455).  psi = -800 -> theta = exp(psi) = 0 for both
    components: R is the all-ones matrix, exactly singular (second pivot 1 - 1 = 0)."""
    from ccgp_amd import api
    D = np.array([[0.1, 0.2], [0.3, 0.8], [0.7, 0.9], [0.4, 0.5]])
    y = np.array([1.0, 2.0, 3.0, 4.0])
    got = R.dot_call("ccgp_R_logpost", R.real(D), R.real([-800.0, -800.0, 0.0]), R.real(y), R.real(1.0),
                     R.integer(api.PRIOR_ISO), R.null(), R.integer(1))
    assert R.is_na(got["val"]).all() and R.is_na(got["beta"]).all() and R.is_na(got["loglik"]).all()
    assert got["R.Inv"].dtype == np.int32 and got["R.Inv"].shape == (1,) and got["R.Inv"][0] == -2 ** 31   # logical NA
    assert R.warnings() == []            # a failed factorisation is a result, not an error


def test_loglik_batch_nan_becomes_na(R, handle):
    from ccgp_amd import api
    D, y, _, _ = load_qian()
    rng = np.random.default_rng(3)
    P = np.array([iso_row(rng.uniform(0.5, 0.9), rng.uniform(0.2, 1), rng.uniform(5, 30), 4) for _ in range(37)])
    got = R.dot_call("ccgp_R_loglik_batch", R.real(D), R.real(y), R.integer(2), R.real(P), R.real(37.0),
                     R.integer(api.MEAN_ZERO_PLUS_TAU2), R.real(2500.0))
    ll, beta, st = handle.loglik_batch(D, y, 2, P, 37.0, api.MEAN_ZERO_PLUS_TAU2, 2500.0)
    assert np.array_equal(got[0], ll) and np.array_equal(got[1], beta) and np.array_equal(got[2], st)
    assert got[2].dtype == np.int32
    # theta = 0 in both components: R = 11', exactly singular -> NA_real_ (not a bare NaN), status = 2, no warning;
    # the other draws of the batch are untouched
    Pz = P[:5].copy()
    Pz[1, 2:] = 0.0
    Pz[4, 2:] = 0.0
    got = R.dot_call("ccgp_R_loglik_batch", R.real(D), R.real(y), R.integer(2), R.real(Pz), R.real(37.0),
                     R.integer(api.MEAN_PROFILE_BETA), R.real(0.0))
    assert list(R.is_na(got[0])) == [False, True, False, False, True] and list(got[2]) == [0, 2, 0, 0, 2]
    assert R.is_na(got[1])[[1, 4]].all() and np.array_equal(got[0][[0, 2, 3]], handle.loglik_batch(D, y, 2, Pz[[0, 2, 3]], 37.0)[0])
    assert R.warnings() == []
    # an invalid call (params with the wrong number of columns -> P != K + K d is not checkable in C; K = 0 is)
    got = R.dot_call("ccgp_R_loglik_batch", R.real(D), R.real(y), R.integer(0), R.real(P[:5]), R.real(37.0),
                     R.integer(0), R.real(0.0))
    assert R.is_na(got[0]).all() and len(R.warnings()) == 1


def test_grid_marginal_is_choose_hyperpars(R, handle):
    D, y, _, _ = load_qian()
    H = load_hyper("hx")[280:300]
    got = R.dot_call("ccgp_R_grid_marginal", R.real(D), R.real(y), R.real(62.0), R.real(H), R.integer(1000), R.real(50.0),
                     R.integer(1), R.real(-1.0))
    vals, arg = handle.grid_marginal(D, y, 62.0, H, 1000, 50.0, True)
    assert np.array_equal(got[0], vals)
    assert int(got[1][0]) == arg + 1 == 13          # which.max is 1-based: row 293 of the full table (HX:774-775)
    # likeli.hyperpars: a single quadruplet as a 1 x 4 matrix, no log (ADV:595)
    one = R.dot_call("ccgp_R_grid_marginal", R.real(D), R.real(y), R.real(62.0), R.real(H[12:13]), R.integer(1000),
                     R.real(50.0), R.integer(0), R.real(-1.0))
    assert one[0].shape == (1,) and one[0][0] == pytest.approx(np.exp(vals[12]), rel=1e-12)
    # a non-positive hyperparameter: warning + NA
    bad = H[:3].copy()
    bad[1, 2] = -1.0
    got = R.dot_call("ccgp_R_grid_marginal", R.real(D), R.real(y), R.real(62.0), R.real(bad), R.integer(100), R.real(50.0),
                     R.integer(1), R.real(-1.0))
    assert R.is_na(got[0]).all() and "must be positive" in R.warnings()[0]


def test_predict_batch_tables(R, handle):
    D, y, Dt, _ = load_gv(50)
    P = np.array([iso_row(0.7, 0.3, 15.0, 9), iso_row(0.9, 0.25, 20.0, 9), iso_row(0.6, 0.4, 12.0, 9)])
    got = R.dot_call("ccgp_R_predict_batch", R.real(D), R.real(y), R.integer(2), R.real(P), R.real(Dt), R.real(10.0))
    mean, var, beta, _ = handle.predict_batch(D, y, 2, P, Dt, 10.0)
    assert got[0].shape == (3, 150) and np.array_equal(got[0], mean) and np.array_equal(got[1], var)
    assert np.array_equal(got[2], beta)


def test_literal_predict_post_factors_beta_sigma2(R, handle):
    """factors (HX:604-613), predict.post's arithmetic (HX:667-670) -> cbind(mean, var), beta.MLE, sigma2.MLE."""
    from ccgp_amd import api
    D, y, Dt, _ = load_qian()
    s2 = 40.0
    lp = handle.logpost(D, y, s2, api.PRIOR_ISO, [np.log(0.3), np.log(15.0), np.log(4.0)])
    Rinv, beta = lp["R_inv"], lp["beta"]
    f = R.dot_call("ccgp_R_factors", R.real(Rinv), R.real(beta), R.real(y))
    want = handle.factors(Rinv, beta, y)
    assert f.shape == (2 * 64 + 1,) and np.array_equal(f, want)
    row = iso_row(0.8, 0.3, 15.0, 4)
    r = handle.mixed_corr_cross(Dt, D, 2, row)
    got = R.dot_call("ccgp_R_predict_from_factors", R.real(r), R.real(beta), R.real(f[:64]), R.real(f[64:128]),
                     R.real(f[128]), R.real(Rinv), R.real(s2))
    m, v = handle.predict_from_factors(r, beta, f[:64], f[64:128], f[128], Rinv, s2)
    assert got.shape == (14, 2) and np.array_equal(got[:, 0], m) and np.array_equal(got[:, 1], v)
    b = R.dot_call("ccgp_R_beta_mle", R.real(Rinv), R.real(y))
    assert b.shape == (1,) and b[0] == handle.beta_mle(Rinv, y)
    s = R.dot_call("ccgp_R_sigma2_mle", R.real(Rinv), R.real(y), R.real(beta))
    assert s[0] == handle.sigma2_mle(Rinv, y, beta)


# ---- the prediction phase of the unchanged scripts: frame -> ONE table (r/ccgp.R's compare.GP / prediction wrappers) ----
def _gv_frame(handle, S=6, size=50, seed=7):
    from ccgp_amd.rsurface import CombinedGP
    D, y, Dt, _ = load_gv(size)
    rng = np.random.default_rng(seed)
    draws = np.column_stack([rng.uniform(0.5, 0.9, S), rng.uniform(0.2, 0.5, S), rng.uniform(10, 20, S)])
    gp = CombinedGP("GV", handle=handle)
    frame = gp.factors_frame_from_draws(draws, D, 10.0, y)           # S x (5 + 2n + n^2), HX:625-644's layout
    P = gp.draws_to_params(D, draws)
    return D, y, Dt, draws, frame, P


def test_prediction_table_from_a_factors_frame_is_predict_batch_bit_for_bit(R, handle):
    """ccgp_R_prediction_table(pars.frame, D.train, D.test, sigma2, y.train, layout, nu): what compare.GP needs for a
    whole test set (HX:713-725, GV:648-676) from the frame factors.frame returned -- as a data frame (list of columns),
    as a numeric matrix, wide (5 + 2n + n^2 columns) or slim (p, theta1, theta2, beta) -- in one ccgp_predict_batch."""
    D, y, Dt, draws, frame, P = _gv_frame(handle)
    S, n = frame.shape[0], D.shape[0]
    assert frame.shape[1] == 5 + 2 * n + n * n
    mean, var, beta, _ = handle.predict_batch(D, y, 2, P, Dt, 10.0)
    args = lambda fr: (fr, R.real(D), R.real(Dt), R.real(10.0), R.real(y), R.integer(0), R.real(0.0))
    for fr in (R.frame(frame), R.real(frame), R.frame(frame[:, :4], names=["p", "theta1", "theta2", "beta"]),
               R.real(frame[:, :3])):
        got = R.dot_call("ccgp_R_prediction_table", *args(fr))
        assert got[0].shape == (S, 150) and np.array_equal(got[0], mean) and np.array_equal(got[1], var)
        assert np.array_equal(got[2], beta)
    # the frame's own beta column is what the device recomputes (factors.frame stores Metro's beta, HX:638)
    assert np.allclose(frame[:, 3], beta, rtol=1e-10, atol=0)
    # a draws table read from a file may carry an integer column
    whole = frame[:, :3].copy()
    whole[:, 2] = np.round(whole[:, 2])
    got = R.dot_call("ccgp_R_prediction_table", *args(R.frame(whole, integer_columns=(2,))))
    want = handle.predict_batch(D, y, 2, np.array([iso_row(r[0], r[1], r[2], 9) for r in whole]), Dt, 10.0)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert R.warnings() == []
    # a draw whose factorisation fails (theta = 0: R = 11'): NA_real_ in its row only
    bad = frame[:, :4].copy()
    bad[2, 1:3] = 0.0
    got = R.dot_call("ccgp_R_prediction_table", *args(R.real(bad)))
    assert R.is_na(got[0][2]).all() and R.is_na(got[1][2]).all() and not R.is_na(got[0][[0, 1, 3, 4, 5]]).any()
    assert np.array_equal(got[0][[0, 1, 3, 4, 5]], mean[[0, 1, 3, 4, 5]])
    # a list whose columns are not equally long is not a frame: warning + NA (no read past the short column)
    ragged = R.frame(frame[:, :3])
    R.L.rmock_list_set(ragged, 2, R.real(frame[:2, 2]))
    got = R.dot_call("ccgp_R_prediction_table", *args(ragged))
    assert R.is_na(got[0]).all() and "do not fit layout" in R.warnings()[-1]
    R.reset()
    # shapes that do not fit the layout: warning + NA, never an error
    got = R.dot_call("ccgp_R_prediction_table", R.real(frame[:, :2]), R.real(D), R.real(Dt), R.real(10.0), R.real(y),
                     R.integer(0), R.real(0.0))
    assert R.is_na(got[0]).all() and "do not fit layout" in R.warnings()[0]


def test_table_cache_lookup_clear_serves_prediction_row_by_row(R, handle):
    """compare.GP's wrapper: ccgp_R_table_cache once, then the script's own apply_pb(D.test, 1, prediction, ...) asks for
    each row of D.test in turn (HX:719); ccgp_R_table_lookup returns the 2 x S block (rows mean, var) that
    apply(pars.frame, 1, predict.post, ...) would have produced (HX:688 transposes it)."""
    D, y, Dt, draws, frame, P = _gv_frame(handle, S=5)
    S = frame.shape[0]
    mean, var, _, _ = handle.predict_batch(D, y, 2, P, Dt, 10.0)
    k = R.dot_call("ccgp_R_table_cache", R.frame(frame[:, :4]), R.real(D), R.real(Dt), R.real(10.0), R.real(y), R.integer(0),
                   R.real(0.0))
    assert k[0] == S
    order = list(range(150)) + [3, 149, 0, 77]           # in order (apply), then out of order (a user's own loop)
    for t in order:
        blk = R.dot_call("ccgp_R_table_lookup", R.real(Dt[t]), R.integer(S))
        assert blk.shape == (2, S) and np.array_equal(blk[0], mean[:, t]) and np.array_equal(blk[1], var[:, t])
    assert R.dot_call("ccgp_R_table_lookup", R.real(Dt[0] + 1e-9), R.integer(S)) is None        # not a row of D.test
    assert R.dot_call("ccgp_R_table_lookup", R.real(Dt[0]), R.integer(S + 1)) is None           # another frame
    assert R.dot_call("ccgp_R_table_lookup", R.real(Dt[0, :5]), R.integer(S)) is None           # another dimension
    assert R.dot_call("ccgp_R_table_clear") is None
    assert R.dot_call("ccgp_R_table_lookup", R.real(Dt[0]), R.integer(S)) is None
    # a failing table call leaves nothing cached
    k = R.dot_call("ccgp_R_table_cache", R.real(frame[:, :2]), R.real(D), R.real(Dt), R.real(10.0), R.real(y), R.integer(0),
                   R.real(0.0))
    assert k[0] == 0 and R.dot_call("ccgp_R_table_lookup", R.real(Dt[0]), R.integer(S)) is None
    assert len(R.warnings()) == 1


def test_predict_post_parses_the_frame_row_in_c(R, handle):
    """predict.post(x.new, D.train, pars, sigma2) HX:655-673 on ONE frame row: r/ccgp.R passes the row as it is, every
    index (beta at 4 / 5, mean.factor, var.factor1, var.factor2, R.Inv: HX:659-663, ANI:611-615) is computed in the shim.
    Same bits as Mixed.corr.vec + the literal arithmetic, and the batched table to rounding."""
    from ccgp_amd.rsurface import CombinedGP
    D, y, Dt, draws, frame, P = _gv_frame(handle, S=3)
    n = D.shape[0]
    mean, var, _, _ = handle.predict_batch(D, y, 2, P, Dt, 10.0)
    for s in range(3):
        row = frame[s]
        got = R.dot_call("ccgp_R_predict_post", R.real(Dt[5]), R.real(D), R.real(row), R.real(10.0), R.integer(0), R.real(0.0))
        assert got.shape == (1, 2)
        r = handle.mixed_corr_cross(Dt[5:6], D, 2, P[s])
        m, v = handle.predict_from_factors(r, row[3], row[4:4 + n], row[4 + n:4 + 2 * n], row[4 + 2 * n],
                                           row[5 + 2 * n:].reshape(n, n, order="F"), 10.0)
        assert got[0, 0] == m[0] and got[0, 1] == v[0]
        assert got[0, 0] == pytest.approx(mean[s, 5], rel=1e-9) and got[0, 1] == pytest.approx(var[s, 5], rel=1e-7)
    # several sites at once: x.new as an m x d matrix -> m rows of cbind(mean, var)
    got = R.dot_call("ccgp_R_predict_post", R.real(Dt[:7]), R.real(D), R.real(frame[0]), R.real(10.0), R.integer(0), R.real(0.0))
    assert got.shape == (7, 2) and np.allclose(got[:, 0], mean[0, :7], rtol=1e-9)
    assert R.warnings() == []
    # the anisotropic script: four leading columns, beta at pars[5] (ANI:604-623)
    Dm = load_maximin(14)
    ym = np.sin(3 * Dm[:, 0]) + Dm[:, 1] ** 2
    gpa = CombinedGP("ANI", handle=handle)
    fa = gpa.factors_frame_from_draws([(0.7, 1.5, 2.5, 4.0)], Dm, 2.0, ym)
    x = np.array([0.1, -0.3])
    got = R.dot_call("ccgp_R_predict_post", R.real(x), R.real(Dm), R.real(fa[0]), R.real(2.0), R.integer(2), R.real(0.0))
    want = gpa.predict_post(x, Dm, fa[0], 2.0)
    assert np.array_equal(got, want)
    tab = R.dot_call("ccgp_R_prediction_table", R.real(fa[:, :4]), R.real(Dm), R.real(x[None]), R.real(2.0), R.real(ym),
                     R.integer(2), R.real(0.0))
    assert tab[0][0, 0] == pytest.approx(got[0, 0], rel=1e-9) and tab[1][0, 0] == pytest.approx(got[0, 1], rel=1e-7)
    # ADV as written: the second scale is theta1 (1 + pars[3]) (ADV:672)
    gpv = CombinedGP("ADV", handle=handle)
    fv = gpv.factors_frame_from_draws([(0.7, 1.5, 2.5)], Dm, 2.0, ym)
    got = R.dot_call("ccgp_R_predict_post", R.real(x), R.real(Dm), R.real(fv[0]), R.real(2.0), R.integer(1), R.real(0.0))
    assert np.array_equal(got, gpv.predict_post(x, Dm, fv[0], 2.0))
    # a slim frame row has no cached terms: warning + NA
    got = R.dot_call("ccgp_R_predict_post", R.real(x), R.real(Dm), R.real(fv[0, :4]), R.real(2.0), R.integer(0), R.real(0.0))
    assert R.is_na(got).all() and "slim" in R.warnings()[-1]


def test_one_dimensional_layouts_select_their_family_for_the_call(R, handle):
    """Layouts 3 / 4: the Matern and Matern + spline families of the 1-D scripts (D1:794-812, D1F:737-754) are selected
    for the duration of the table call and the handle is Gaussian again afterwards."""
    from ccgp_amd import api
    X = np.linspace(0.02, 0.98, 12)[:, None]
    y = np.sin(6 * X[:, 0]) + 0.3 * X[:, 0]
    Xt = np.linspace(0.1, 0.9, 5)[:, None]
    draws = np.array([[0.7, 0.4, 0.15], [0.6, 0.5, 0.2]])
    P = np.array([[r[0], 1 - r[0], r[1], r[2]] for r in draws])
    for layout, fam in ((3, api.KERNEL_MATERN), (4, api.KERNEL_MATERN_SPLINE)):
        got = R.dot_call("ccgp_R_prediction_table", R.real(draws), R.real(X), R.real(Xt), R.real(1.5), R.real(y),
                         R.integer(layout), R.real(5.0))
        handle.set_kernel(fam, 5.0)
        try:
            want = handle.predict_batch(X, y, 2, P, Xt, 1.5)
        finally:
            handle.set_kernel(api.KERNEL_GAUSS)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        # back to Gaussian for the next call
        assert np.array_equal(R.dot_call("ccgp_R_corr_matrix", R.real(X), R.real([0.4])), handle.corr_matrix(X, [0.4]))
    assert R.warnings() == []


def test_mixed_logdet_designs(R, handle):
    rng = np.random.default_rng(11)
    designs = rng.random((6, 14, 2))
    row = iso_row(0.8, 1.0, 12.0, 2)
    Xs = np.stack([np.asfortranarray(Dd).ravel(order="F") for Dd in designs], axis=1)      # n*d x B, as r/ccgp.R passes it
    got = R.dot_call("ccgp_R_mixed_logdet_designs", R.real(Xs), R.integer(14), R.integer(2), R.integer(2), R.real(row))
    want, _ = handle.mixed_logdet_designs(designs, 2, row)
    assert np.array_equal(got, want)


def test_set_kernel_switches_the_family_for_later_calls(R, handle):
    from ccgp_amd import api
    X = np.linspace(0.05, 0.95, 8)[:, None]
    rc = R.dot_call("ccgp_R_set_kernel", R.integer(api.KERNEL_MATERN), R.real(5.0))
    assert rc[0] == 0
    got = R.dot_call("ccgp_R_corr_matrix", R.real(X), R.real([0.4]))
    handle.set_kernel(api.KERNEL_MATERN, 5.0)
    try:
        want = handle.corr_matrix(X, [0.4])
    finally:
        handle.set_kernel(api.KERNEL_GAUSS)
    assert np.array_equal(got, want)
    assert R.dot_call("ccgp_R_set_kernel", R.integer(0), R.real(0.0))[0] == 0
    assert np.array_equal(R.dot_call("ccgp_R_corr_matrix", R.real(X), R.real([0.4])), handle.corr_matrix(X, [0.4]))
    # nu outside the validated range: return code + warning, family unchanged
    assert R.dot_call("ccgp_R_set_kernel", R.integer(1), R.real(50.0))[0] == -1 and len(R.warnings()) == 1


def test_dispatch_goes_through_the_registration(R):
    src = open(os.path.join(ROOT, "r", "ccgp_shim.c")).read()
    import re
    table = dict((n, int(k)) for n, k in re.findall(r'\{"(ccgp_R_\w+)", \(DL_FUNC\)&\w+, (\d+)\}', src))
    assert R.routines == table and R.L.rmock_dynamic_symbols() == 0
    with pytest.raises(rmock.RError, match="Incorrect number of arguments"):
        R.dot_call("ccgp_R_beta_mle", R.real(np.eye(2)))
    with pytest.raises(rmock.RError, match="not available"):
        R.dot_call("ccgp_loglik_batch")
    assert R.dot_call("ccgp_R_devices")[0] == 1


def test_ccgp_devices_routes_the_batched_calls_through_ccgp_multi(R, handle):
    """CCGP_DEVICES: the shim shards the batched calls over several devices (one-GPU box: device 0 listed twice --
    shards then share it; distinct devices are the driver's 8-GPU run).  Same bits as the single handle."""
    from ccgp_amd import api
    D, y, Dt, _ = load_gv(90)
    rng = np.random.default_rng(5)
    P = np.array([iso_row(rng.uniform(0.5, 0.9), rng.uniform(0.2, 0.5), rng.uniform(10, 20), 9) for _ in range(11)])
    H = load_hyper("hx")[:7]
    Dq, yq, _, _ = load_qian()
    R.unload()
    os.environ["CCGP_DEVICES"] = "0,0"
    try:
        R.L.rmock_load()
        assert R.dot_call("ccgp_R_devices")[0] == 2
        got = R.dot_call("ccgp_R_loglik_batch", R.real(D), R.real(y), R.integer(2), R.real(P), R.real(10.0), R.integer(0),
                         R.real(0.0))
        ll, beta, st = handle.loglik_batch(D, y, 2, P, 10.0)
        assert np.array_equal(got[0], ll) and np.array_equal(got[1], beta)
        got = R.dot_call("ccgp_R_predict_batch", R.real(D), R.real(y), R.integer(2), R.real(P), R.real(Dt[:20]), R.real(10.0))
        mean, var, b2, _ = handle.predict_batch(D, y, 2, P, Dt[:20], 10.0)
        assert np.array_equal(got[0], mean) and np.array_equal(got[1], var) and np.array_equal(got[2], b2)
        # the frame-driven table of compare.GP's wrapper is sharded over the draws as well
        fr = np.column_stack([P[:, 0], P[:, 2], P[:, 11]])                  # (p, theta1, theta2) back from the rows
        got = R.dot_call("ccgp_R_prediction_table", R.real(fr), R.real(D), R.real(Dt[:20]), R.real(10.0), R.real(y),
                         R.integer(0), R.real(0.0))
        assert np.array_equal(got[0], mean) and np.array_equal(got[1], var) and np.array_equal(got[2], b2)
        got = R.dot_call("ccgp_R_grid_marginal", R.real(Dq), R.real(yq), R.real(62.0), R.real(H), R.integer(200), R.real(50.0),
                         R.integer(1), R.real(-1.0))
        vals, arg = handle.grid_marginal(Dq, yq, 62.0, H, 200, 50.0, True)
        assert np.array_equal(got[0], vals) and got[1][0] == arg + 1
        # the family switch reaches every shard
        assert R.dot_call("ccgp_R_set_kernel", R.integer(0), R.real(0.0))[0] == 0
        assert R.warnings() == []
        # a device that does not exist: warning, and the calls fall back to the single handle
        R.unload()
        os.environ["CCGP_DEVICES"] = "0,63"
        R.L.rmock_load()
        assert R.dot_call("ccgp_R_devices")[0] == 1
        assert any("could not be opened" in w for w in R.warnings())
    finally:
        R.unload()
        os.environ.pop("CCGP_DEVICES", None)
        R.L.rmock_load()


@pytest.mark.parametrize("n,d,S,m,seed", [(9, 1, 3, 1, 1), (64, 4, 5, 33, 2), (100, 2, 2, 150, 3), (129, 3, 4, 7, 4),
                                           (300, 5, 3, 131, 5), (520, 2, 2, 64, 6)])
def test_batched_routines_at_random_shapes_including_the_blocked_path(R, handle, n, d, S, m, seed):
    """The shim's batched routines hand R's column-major buffers straight through; nothing in them depends on the size of the
    design, so they must serve n > 128 (the blocked sweep behind the same C entry points) as they serve the scripts' own
    sizes: likelihood batch, prediction tables from a parameter matrix and from a slim frame, logpost with R.Inv, design
    log-determinants -- each equal to api.Handle (ctypes -> the same C ABI) bit for bit."""
    rng = np.random.default_rng(seed)
    X, y = synthetic_design(n, d, seed)
    s = n ** (2.0 / d) / d
    draws = np.column_stack([rng.uniform(0.55, 0.9, S), rng.uniform(0.05, 0.3, S) * s, rng.uniform(2.0, 4.0, S) * s])
    P = np.array([iso_row(r[0], r[1], r[2], d) for r in draws])
    Xt = rng.random((m, d))
    got = R.dot_call("ccgp_R_loglik_batch", R.real(X), R.real(y), R.integer(2), R.real(P), R.real(1.4), R.integer(0), R.real(0.0))
    ll, beta, st = handle.loglik_batch(X, y, 2, P, 1.4)
    assert not st.any() and np.array_equal(got[0], ll) and np.array_equal(got[1], beta)
    got = R.dot_call("ccgp_R_predict_batch", R.real(X), R.real(y), R.integer(2), R.real(P), R.real(Xt), R.real(1.4))
    mean, var, beta2, _ = handle.predict_batch(X, y, 2, P, Xt, 1.4)
    assert got[0].shape == (S, m) and np.array_equal(got[0], mean) and np.array_equal(got[1], var) and np.array_equal(got[2], beta2)
    slim = np.column_stack([draws, beta2])
    got = R.dot_call("ccgp_R_prediction_table", R.frame(slim, names=["p", "theta1", "theta2", "beta"]), R.real(X), R.real(Xt),
                     R.real(1.4), R.real(y), R.integer(0), R.real(0.0))
    assert np.array_equal(got[0], mean) and np.array_equal(got[1], var)
    t = np.array([np.log(draws[0, 1]), np.log(draws[0, 2]), np.log(draws[0, 0] / (1 - draws[0, 0]))])
    from ccgp_amd import api
    got = R.dot_call("ccgp_R_logpost", R.real(X), R.real(t), R.real(y), R.real(1.4), R.integer(api.PRIOR_GV), R.null(), R.integer(1))
    want = handle.logpost(X, y, 1.4, api.PRIOR_GV, t, None, True)
    assert got["val"][0] == want["val"] and got["beta"][0] == want["beta"] and np.array_equal(got["R.Inv"], want["R_inv"])
    designs = np.stack([synthetic_design(n, d, seed + 10 + i)[0] for i in range(3)])
    Xs = np.stack([np.asfortranarray(Dd).ravel(order="F") for Dd in designs], axis=1)
    got = R.dot_call("ccgp_R_mixed_logdet_designs", R.real(Xs), R.integer(n), R.integer(d), R.integer(2), R.real(P[0]))
    want_ld, _ = handle.mixed_logdet_designs(designs, 2, P[0])
    assert np.array_equal(got, want_ld)
    assert R.warnings() == []


# ------------------------------------------------------------------------------- Metro on the batched path
@pytest.mark.parametrize("script,prior,q", [("GV", 1, 3), ("HX", 0, 3), ("ANI", 3, 4)])
def test_metro_steps_is_the_sequential_chain(R, handle, script, prior, q):
    """ccgp_R_metro_steps (the next m iterations of Metro's loop, HX:505-535, as ONE device call) against the chain the
    script runs: one logpost per proposal (Handle.logpost = ccgp_logpost), accept on l.cand - l.old > log(u).  Same
    pre-drawn (u, innovation) pairs for both: accepted flags, states, values and betas must agree BIT FOR BIT over 48
    proposals in blocks of m = 1, 3, 4 and 6 -- and with the tree walk of fit.Metro(speculate = m), which reads the same
    ccgp_logpost_batch values."""
    from ccgp_amd import api
    rng = np.random.default_rng(2014 + prior)
    if script == "GV":
        D, y, _, _ = load_gv(50, 1)
        start, s2, pars = np.array([np.log(0.3), np.log(15.0), 0.8]), 10.25, None
    elif script == "HX":
        D, y, _, _ = load_qian()
        start, s2, pars = np.array([np.log(0.3), np.log(15.0), 0.8]), 64.0, np.array([7.0, 3.0, 3.0, 28.0])
    else:
        D = load_maximin(100)[:60]
        y = (np.sin(2 * D[:, 0]) + np.cos(4 * D[:, 0])) * (np.sin(8 * D[:, 1]) + np.cos(4 * D[:, 1]))
        start, s2, pars = np.array([np.log(2.0), np.log(3.0), 0.5, np.log(4.0)]), float(np.var(y, ddof=1)), None
    U = np.linalg.cholesky(np.diag(np.full(q, 0.25)) + 0.02).T        # any proposal factor: chol(sqrt(2) V) in the script
    T = 48
    u = rng.random(T)
    E = rng.normal(size=(T, q)) @ U
    E[17] = 60.0                                                       # one proposal far outside: exp() overflow -> NaN / reject

    def one(theta):
        r = handle.logpost(D, y, s2, prior, theta, pars, want_Rinv=False)
        return (r["val"] if r["status"] == 0 else np.nan), r["beta"]

    # the script's loop, one logpost per proposal
    th, (lo, bo) = start.copy(), one(start)
    seq = []
    for t in range(T):
        cand = th + E[t]
        vc, bc = one(cand)
        a = bool(np.isfinite(vc) and (vc - lo) > np.log(u[t]))
        if a:
            th, lo, bo = cand, vc, bc
        seq.append((a, th.copy(), lo, bo))
    assert 5 < sum(s[0] for s in seq) < T - 5                          # a chain that both accepts and rejects

    prior_vec = np.concatenate([[float(prior)], pars]) if pars is not None else np.array([float(prior)])
    for m in (1, 3, 4, 6):
        th, (lo, bo) = start.copy(), one(start)
        t = 0
        while t < T:
            mm = min(m, T - t)
            r = R.dot_call("ccgp_R_metro_steps", R.real(D), R.real(y), R.real(np.array([s2])), R.real(prior_vec), R.real(th),
                           R.real(np.array([lo, bo])), R.real(u[t:t + mm]), R.real(E[t:t + mm].reshape(mm, q)))
            acc, theta, val, beta, nev = r["accepted"], r["theta"], r["val"], r["beta"], r["evaluated"]
            assert int(np.ravel(nev)[0]) == 2 ** mm - 1
            for i in range(mm):
                a, th_s, lo_s, bo_s = seq[t + i]
                assert bool(acc[i]) == a, (m, t + i)
                assert np.array_equal(np.atleast_2d(theta)[i], th_s) and val[i] == lo_s and beta[i] == bo_s, (m, t + i)
            th, lo, bo = np.atleast_2d(theta)[mm - 1].copy(), float(val[mm - 1]), float(beta[mm - 1])
            t += mm
        assert R.warnings() == [] or all("libccgp" not in w for w in R.warnings())
    assert R.is_na(np.atleast_1d(R.dot_call("ccgp_R_metro_steps", R.real(D), R.real(y), R.real(np.array([s2])), R.real(prior_vec),
                                            R.real(start), R.real(np.array(one(start))), R.real(u[17:18]),
                                            R.real(E[17:18].reshape(1, q)))["cand.val"]))[0]

    # the same values through the Python host layer's batch (fit.logpost_batch -> ccgp_logpost_batch)
    vb, bb, _, _ = handle.logpost_batch(D, y, s2, prior, np.stack([start + E[0], start + E[1]]), pars)
    assert vb[0] == one(start + E[0])[0] and bb[1] == one(start + E[1])[1]
