"""The compiled CPU baseline evaluator (oracle/cpu_baseline, bench.py's `cpu_baseline` leg) computes what the
oracle computes -- otherwise its timing would be a timing of something else.  CPU only."""
import numpy as np
import pytest

from conftest import golden, load_gv, load_qian, synthetic_design
from oracle import ccgp_oracle as orc
from oracle.cpu_baseline import loader as cpu


def test_loading_the_evaluator_leaves_denormals_alone():
    """Round-2 advisor finding: built with -ffast-math the library linked crtfastmath.o, and dlopen set FTZ/DAZ for
    the whole process -- every numpy / scipy oracle computation after it ran with denormals flushed to zero."""
    cpu.load()
    assert np.float64(1e-310) * 0.5 != 0.0
    assert np.exp(np.float64(-720.0)) != 0.0


def test_lapack_is_bound_from_scipy_openblas():
    assert cpu.lapack_bound(), "scipy's OpenBLAS not found: the baseline would time the plain C Cholesky"


@pytest.mark.parametrize("mode", [0, 1])
def test_loglik_matches_the_oracle_on_qian(mode):
    D, y, _, _ = load_qian()
    g = golden("hx_golden.json")
    draws = np.asarray(g["draws"])[:6]
    P = np.array([orc.params_from_iso(p, t1, t2, 4) for p, t1, t2 in draws])
    s2, tau2 = 37.0, 2500.0 if mode else 0.0
    ll, beta, st = cpu.loglik_batch(D, y, 2, P, s2, mode, tau2, threads=2)
    assert not st.any()
    for b in range(P.shape[0]):
        w, Th = orc.unpack_params(P[b], 2, 4)
        wl, wb = orc.loglik_general(D, y, w, Th, s2, mode, tau2)
        assert ll[b] == pytest.approx(wl, rel=1e-9)
        if mode == 0:
            assert beta[b] == pytest.approx(wb, rel=1e-9)


def test_loglik_general_K3_anisotropic():
    X, y = synthetic_design(150, 5, 3)
    rng = np.random.default_rng(1)
    P = np.array([np.concatenate([[0.5, 0.3, 0.2], np.exp(rng.uniform(-0.5, 3.5, 15))]) for _ in range(3)])
    P[:, -5:] = np.maximum(P[:, -5:], 25.0)
    ll, beta, st = cpu.loglik_batch(X, y, 3, P, 1.0, 0, 0.0, threads=3)
    for b in range(3):
        w, Th = orc.unpack_params(P[b], 3, 5)
        wl, wb = orc.loglik_general(X, y, w, Th, 1.0)
        assert ll[b] == pytest.approx(wl, rel=1e-8) and beta[b] == pytest.approx(wb, rel=1e-8)


def test_predict_matches_the_oracle_on_ground_vibrations():
    D, y, Dt, _ = load_gv(50)
    draws = np.array([[0.7, 0.3, 15.0], [0.9, 0.25, 20.0]])
    P = np.array([orc.params_from_iso(p, t1, t2, 9) for p, t1, t2 in draws])
    mean, var = cpu.predict_batch(D, y, 2, P, Dt[:7], 10.0, threads=2)
    wm, wv, _ = orc.predict_table(D, y, draws, Dt[:7], 10.0)
    np.testing.assert_allclose(mean, wm, rtol=1e-8)
    np.testing.assert_allclose(var, wv, rtol=1e-6, atol=1e-9)


def test_non_positive_definite_gives_nan_and_status():
    X = np.array([[0.1], [0.1], [0.5]])          # duplicated site: singular R
    ll, _, st = cpu.loglik_batch(X, np.array([1.0, 2.0, 3.0]), 1, np.array([[1.0, 2.0]]), 1.0)
    assert st[0] != 0 and np.isnan(ll[0])
