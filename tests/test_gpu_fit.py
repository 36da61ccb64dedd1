"""End-to-end fit/predict on the device path, checked STATISTICALLY against the only output the
reference records (Ground Vibrations Emulator/Results/Size 50 Results 1.txt: RMSPE 2.722,
95 % interval coverage 0.973, mean predictive quantile 0.503 -- one unseeded R run)."""
import numpy as np
import pytest

from oracle import ccgp_oracle as orc

from conftest import load_gv, load_qian

pytestmark = pytest.mark.gpu


def test_ground_vibrations_fit_reproduces_the_recorded_run_statistically(handle):
    # (the per-test-point comparison with the recorded table is tests/test_reference_pins_gpu.py)
    from ccgp_amd import fit
    from ccgp_amd.rsurface import CombinedGP
    D, y, Dt, yt = load_gv(50)
    gp = CombinedGP("GV", handle=handle)
    sigma2, theta, beta = fit.ordinary_kriging_sigma2(handle, D, y)
    assert 0.1 * np.var(y) < sigma2 < 20 * np.var(y) and np.all(theta > 0)
    # driver block GV:688-695: start c(1,1,0), N.max 5000, samp.size 1000, alpha.geweke 0.5, batch 20
    table = fit.Combined_GP_fit(gp, D, y, Dt, [1.0, 1.0, 0.0], 5000, 1000, 0.5, 20, alpha=0.05, y_new=yt,
                                sigma2=sigma2, rng=20140101)
    s = fit.comparison_summary(table)
    print("GV size-50 sample 1: RMSPE %.3f (reference run 2.722), coverage %.3f (0.973), mean quantile %.3f (0.503), "
          "accepted %d of %d proposals" % (s["rmspe"], s["coverage"], s["mean_quantile"], table["chain"]["accepted"],
                                           table["chain"]["proposals"]))
    assert 2.2 < s["rmspe"] < 3.3
    assert s["coverage"] >= 0.88
    assert 0.40 < s["mean_quantile"] < 0.60
    draws = table["draws"]
    assert draws.shape == (1000, 3) and np.all((draws[:, 0] > 0) & (draws[:, 0] < 1)) and np.all(draws[:, 1:] > 0)


def test_heat_exchanger_fit(handle):
    from ccgp_amd import fit
    from ccgp_amd.rsurface import CombinedGP
    D, y, Dt, yt = load_qian()
    gp = CombinedGP("HX", handle=handle)
    sigma2, _, _ = fit.ordinary_kriging_sigma2(handle, D, y)
    # driver block HX:736-742, 774-775: start c(1,2.7,0), priors (7,3), (3,28)
    table = fit.Combined_GP_fit(gp, D, y, Dt, [1.0, 2.7, 0.0], 5000, 1000, 0.5, 20, y_new=yt, sigma2=sigma2,
                                theta1_pars=(7, 3), theta2_pars=(3, 28), rng=7)
    s = fit.comparison_summary(table)
    print("Qian: RMSPE %.3f, coverage %.2f" % (s["rmspe"], s["coverage"]))
    assert s["rmspe"] < 0.5 * np.std(yt)          # far better than predicting the mean
    assert s["coverage"] >= 0.7                   # 14 test points only


def test_speculative_metro_on_the_device_is_the_sequential_chain(handle):
    """Metro on the BATCHED device path.  speculate=5 sends 31 candidates per round
    trip; every candidate is evaluated by its own wave of the same kernel, so the chain is the
    sequential one bit for bit -- with a fifth of the device round trips."""
    import time
    from ccgp_amd import fit
    from ccgp_amd.rsurface import CombinedGP
    D, y, _, _ = load_qian()
    gp = CombinedGP("HX", handle=handle)
    s2 = float(np.var(y, ddof=1))
    kw = dict(theta1_pars=(7, 3), theta2_pars=(3, 28))
    args = (gp, [1.0, 2.7, 0.0], 600, 200, 20, 0.5, D, s2, y)
    t0 = time.perf_counter()
    seq = fit.Metro(*args, rng=np.random.default_rng(3), **kw)
    t1 = time.perf_counter()
    spec = fit.Metro(*args, rng=np.random.default_rng(3), speculate=5, **kw)
    t2 = time.perf_counter()
    np.testing.assert_array_equal(seq["sample"], spec["sample"])
    np.testing.assert_array_equal(seq["beta"], spec["beta"])
    assert (seq["accepted"], seq["proposals"]) == (spec["accepted"], spec["proposals"])
    assert spec["device_batches"] * 4 < seq["device_batches"]
    print("Metro on Qian: %d proposals, sequential %d round trips %.2f s, speculate=5 %d round trips %.2f s"
          % (seq["proposals"], seq["device_batches"], t1 - t0, spec["device_batches"], t2 - t1))


def test_one_dimensional_fit_matern(handle):
    """The 1-D script end to end on the device (BASELINE config 1's surface): driver block D1:1078-1099
    scaled down -- n.train = 8 points, Matern nu = 5, simulator f2(x) = sin(10 x) (D1:331-339), 50 test
    sites on [0, 1] -- through laplace + Metro(speculate) + prediction with the Matern family."""
    from ccgp_amd import fit
    from ccgp_amd.rsurface import CombinedGP1D
    X = (np.arange(8) + np.array([0.3, 0.7, 0.5, 0.2, 0.8, 0.4, 0.6, 0.5])) / 8.0       # a Latin hypercube in [0, 1]
    f = lambda x: np.sin(10.0 * x)
    D, y = X.reshape(-1, 1), f(X)
    Dn = np.linspace(0.0, 1.0, 50).reshape(-1, 1)
    gp = CombinedGP1D(5.0, handle=handle)
    table = fit.Combined_GP_fit(gp, D, y, Dn, [0.0, 1.5, 0.0], 3000, 600, 0.5, 20, net_samp_size=300, y_new=f(Dn[:, 0]),
                                sigma2=float(np.var(y, ddof=1)), rng=5, speculate=4)
    s = fit.comparison_summary(table)
    print("1-D Matern fit: RMSPE %.3f, coverage %.2f, %d proposals in %d device round trips"
          % (s["rmspe"], s["coverage"], table["chain"]["proposals"], table["chain"]["device_batches"]))
    assert s["rmspe"] < 0.35 * np.std(f(Dn[:, 0]))
    assert s["coverage"] >= 0.8
    # interpolation at the training sites: the predictive mean reproduces y.train, the variance collapses
    t = gp.prediction_table(X, table["draws"][:5], X, float(np.var(y, ddof=1)), y)
    np.testing.assert_allclose(t["mean"], np.tile(y, (5, 1)), atol=1e-6)
    assert np.all(np.abs(t["var"]) < 1e-6)


def test_matern_mles_is_the_minimum_of_the_1d_scripts_log_likeli(handle):
    """MLEs(D, y, nu) (D1:455-471): theta minimises log.likeli = log det R + n log sigma2.MLE (D1:437-444); beta and
    sigma2 are beta.MLE / sigma2.MLE at that theta.  Checked against the oracle's literal log.likeli: the device's
    optimum is a local minimum of it, no point of a dense scan lies below it, and sigma2 / beta agree."""
    from ccgp_amd import fit
    from conftest import golden
    g = golden("d1_golden.json")
    X, y, nu = np.array(g["X"]), np.array(g["y"]), g["nu"]
    m = fit.matern_MLEs(handle, X, y, nu)
    f0 = orc.log_likeli_1d(nu, m["theta"], X.reshape(-1, 1), y)
    scan = []
    for t in np.exp(np.linspace(np.log(0.02), np.log(20.0), 200)):
        try:
            scan.append(orc.log_likeli_1d(nu, t, X.reshape(-1, 1), y))
        except np.linalg.LinAlgError:            # solve(R, tol = 1e-16) refuses the matrix: the reference retries
            pass
    assert len(scan) > 50 and f0 <= min(scan) + 1e-8
    for eps in (1e-3, -1e-3):
        assert orc.log_likeli_1d(nu, m["theta"] * (1 + eps), X.reshape(-1, 1), y) >= f0 - 1e-9
    R_inv = orc.solve_inverse(orc.corr_matrix_matern(nu, X.reshape(-1, 1), m["theta"]), tol=1e-16)
    beta = orc.beta_mle(R_inv, y)
    assert m["beta"] == pytest.approx(beta, rel=1e-7)
    assert m["sigma2"] == pytest.approx(orc.sigma2_mle(R_inv, y, beta), rel=1e-7)


def test_one_dimensional_fit_takes_sigma2_from_its_own_mles(handle):
    """Combined.GP.fit of the 1-D script (D1:989-1001): no sigma2 argument -- it comes from MLEs()."""
    from ccgp_amd import fit
    from ccgp_amd.rsurface import CombinedGP1D
    X = (np.arange(8) + np.array([0.3, 0.7, 0.5, 0.2, 0.8, 0.4, 0.6, 0.5])) / 8.0
    f = lambda x: np.sin(10.0 * x)
    D, y = X.reshape(-1, 1), f(X)
    Dn = np.linspace(0.0, 1.0, 40).reshape(-1, 1)
    gp = CombinedGP1D(5.0, handle=handle)
    table = fit.Combined_GP_fit(gp, D, y, Dn, [0.0, 1.5, 0.0], 2000, 400, 0.5, 20, net_samp_size=200, y_new=f(Dn[:, 0]),
                                rng=5, speculate=4)
    assert table["sigma2"] == pytest.approx(fit.matern_MLEs(handle, D, y, 5.0)["sigma2"], rel=1e-12)
    s = fit.comparison_summary(table)
    assert s["rmspe"] < 0.35 * np.std(f(Dn[:, 0])) and s["coverage"] >= 0.8


def test_anisotropic_2d_fit(handle):
    """The 2-D anisotropic script end to end (driver block ANI:779-804 scaled down): maximin-14 design,
    simulator f3 (ANI:337), four transformed parameters (psi1, psi2, phi, zeta), prior ANI:462."""
    from ccgp_amd import fit
    from ccgp_amd.rsurface import CombinedGP
    from conftest import load_maximin
    from oracle import ccgp_oracle as orc
    D = load_maximin(14)
    y = np.array([orc.test_function_2d(a, b, 3) for a, b in D])
    g = np.linspace(0.05, 0.95, 6)
    Dn = np.array([[a, b] for a in g for b in g])
    yn = np.array([orc.test_function_2d(a, b, 3) for a, b in Dn])
    gp = CombinedGP("ANI", handle=handle)
    table = fit.Combined_GP_fit(gp, D, y, Dn, [0.0, 0.0, 0.0, 0.0], 3000, 500, 0.5, 20, net_samp_size=250, y_new=yn,
                                sigma2=float(np.var(y, ddof=1)), rng=11, speculate=4)
    s = fit.comparison_summary(table)
    print("2-D anisotropic fit on maximin-14: RMSPE %.3f (sd of y %.3f), coverage %.2f" % (s["rmspe"], np.std(yn), s["coverage"]))
    assert table["draws"].shape == (250, 4) and np.all(table["draws"][:, 1:] > 0)
    assert s["rmspe"] < 0.6 * np.std(yn)
    assert s["coverage"] >= 0.75
