"""The ORACLE pinned to what the reference itself holds (CPU).

The reference ships no tests and no golden vectors; the one output file it records is
`Ground Vibrations Emulator/Results/Size 50 Results 1.txt` (written by GV:759-761).  Its
`y.hat.single / LL.single / UL.single` columns are deterministic functions of mlegp's fitted ordinary-
kriging model, whose parameters `tests/golden/recover_mlegp_gv.py` reads back from those same columns to
rounding level (10 free parameters against 300 recorded numbers).  With them, the recorded columns are a
known-answer test for the reference arithmetic the oracle restates: corr.matrix / corr.vec for general d
(GV:327 = HX:328-337, HX:367-375), solve (HX:454), beta.MLE (HX:384-388), the predictive mean
beta + r' R^-1 (y - beta) and the quadratic form r' R^-1 r of predict.post (HX:667-670)."""
import os

import numpy as np
from scipy.stats import t as student_t

from conftest import DATA, golden, load_gv
from ccgp_amd.tables import read_table
from oracle import ccgp_oracle as orc


def recorded_table():
    names, res = read_table(os.path.join(DATA, "gv", "results_50_1.txt"))
    return {n: res[:, i] for i, n in enumerate(names)}, res[:, :9]


def test_recorded_table_is_the_size50_sample1_test_set():
    rec, Dt = recorded_table()
    _, _, Dtest, ytest = load_gv(50)
    np.testing.assert_array_equal(Dt, Dtest)
    np.testing.assert_array_equal(rec["y.true"], ytest)


def test_oracle_reproduces_the_recorded_single_gp_columns():
    fx = golden("gv_mlegp_recovered.json")
    rec, Dt = recorded_table()
    D, y, _, _ = load_gv(50)
    theta = np.array(fx["theta"])
    R_inv = orc.solve_inverse(orc.corr_matrix(D, theta))
    beta = orc.beta_mle(R_inv, y)
    assert abs(beta - fx["beta"]) < 1e-9
    mf, v1, v2 = orc.factors(R_inv, beta, y)
    qt = student_t.ppf(0.975, D.shape[0] - 1)          # GV:664-665: qt(1 - alpha/2, df = n.train - 1)
    for t in range(Dt.shape[0]):
        r = orc.corr_vec(Dt[t], D, theta)
        mean, var_full = orc.predict_post_from_factors(r, beta, mf, v1, v2, R_inv, fx["sigma2"])
        assert abs(mean - rec["y.hat.single"][t]) < 1e-8
        # mlegp's se.fit^2 = sigma2 (1 - r' R^-1 r): predict.post's variance (HX:669) minus its beta-uncertainty term
        se2 = var_full - fx["sigma2"] * (1.0 - v1 @ r) ** 2 / v2
        se = np.sqrt(se2)
        assert abs(mean - se * qt - rec["LL.single"][t]) < 1e-8
        assert abs(mean + se * qt - rec["UL.single"][t]) < 1e-8
    # derived figure quoted in SURVEY.md section 4 for this column
    assert abs(np.sqrt(np.mean((rec["y.hat.single"] - rec["y.true"]) ** 2)) - 2.687) < 1e-3


def test_geweke_window_is_samp_size_long():
    """HX:530 tests samp[(k-samp.size):(k-1)]: exactly samp.size accepted draws."""
    from ccgp_amd import fit
    samp = np.arange(40.0).reshape(20, 2)
    w = fit.geweke_window(samp, 15, 10)
    assert w.shape == (10,) and w[0] == samp[5, 0] and w[-1] == samp[14, 0]
