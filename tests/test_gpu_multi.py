"""Several devices behind ONE host process (ccgp_multi_*, include/ccgp.h): a multi handle on one device, and a
rehearsal with two / three shards that share device 0, return bit-identical results to the single handle."""
import numpy as np
import pytest

from conftest import golden, load_gv, load_hyper, load_qian, synthetic_design
from oracle import ccgp_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[[0], [0, 0], [0, 0, 0]], ids=lambda d: "%d-shard" % len(d))
def multi(request):
    from ccgp_amd import api
    m = api.MultiHandle(request.param)
    yield m
    m.close()


def test_loglik_small_and_blocked(handle, multi):
    D, y, _, _ = load_qian()
    draws = np.asarray(golden("hx_golden.json")["draws"])
    P = np.array([orc.params_from_iso(p, t1, t2, 4) for p, t1, t2 in draws])[:10]     # 10 rows: ragged shards
    for mode, tau2 in ((0, 0.0), (1, 2500.0)):
        a = handle.loglik_batch(D, y, 2, P, 37.0, mode, tau2)
        b = multi.loglik_batch(D, y, 2, P, 37.0, mode, tau2)
        for u, v in zip(a, b):
            np.testing.assert_array_equal(u, v)
    X, yy = synthetic_design(300, 3, 1)                                                 # blocked path: 3 x 3 tiles
    rows = np.array([np.concatenate([[0.5, 0.5], [1.0, 2.0, 3.0 + k], [30.0, 40.0, 50.0]]) for k in range(5)])
    a = handle.loglik_batch(X, yy, 2, rows, 1.0)
    b = multi.loglik_batch(X, yy, 2, rows, 1.0)
    for u, v in zip(a, b):
        np.testing.assert_array_equal(u, v)
    one = multi.loglik_batch(X, yy, 2, rows[:1], 1.0)                                   # fewer rows than shards
    np.testing.assert_array_equal(one[0], a[0][:1])


def test_grid_sharded_by_row(handle, multi):
    D, y, _, _ = load_qian()
    H = load_hyper("hx")[:21]
    va, aa, la = handle.grid_marginal(D, y, 37.0, H, 200, 50.0, True, want_logs=True)
    vb, ab, lb = multi.grid_marginal(D, y, 37.0, H, 200, 50.0, True, want_logs=True)
    np.testing.assert_array_equal(va, vb)
    np.testing.assert_array_equal(la, lb)
    assert aa == ab


def test_predict_sharded_by_draw(handle, multi):
    D, y, Dt, _ = load_gv(50)
    draws = np.array([[0.6 + 0.04 * s, 0.25 + 0.01 * s, 14.0 + s] for s in range(7)])
    P = np.array([orc.params_from_iso(p, t1, t2, 9) for p, t1, t2 in draws])
    a = handle.predict_batch(D, y, 2, P, Dt, 10.0)
    b = multi.predict_batch(D, y, 2, P, Dt, 10.0)
    for u, v in zip(a, b):
        np.testing.assert_array_equal(u, v)


def test_errors_come_back_with_the_shard_message(multi):
    from ccgp_amd import api
    D, y, _, _ = load_qian()
    with pytest.raises(api.CcgpError):
        multi.loglik_batch(D, y, 2, np.ones((4, 10)), 1.0, mean_mode=7)


def test_kernel_family_reaches_every_shard(handle, multi):
    """ccgp_multi_set_kernel: the 1-D script's Matern family on every shard's handle."""
    from ccgp_amd import api
    X = np.linspace(0.05, 0.95, 8)[:, None]
    y = np.sin(10.0 * X[:, 0])
    P = np.array([[0.6 + 0.03 * b, 0.4 - 0.03 * b, 0.3 + 0.02 * b, 0.08 + 0.005 * b] for b in range(7)])
    try:
        handle.set_kernel(api.KERNEL_MATERN, 5.0)
        multi.set_kernel(api.KERNEL_MATERN, 5.0)
        a = handle.loglik_batch(X, y, 2, P, 1.0)
        b = multi.loglik_batch(X, y, 2, P, 1.0)
    finally:
        handle.set_kernel(api.KERNEL_GAUSS, 0.0)
        multi.set_kernel(api.KERNEL_GAUSS, 0.0)
    for u, v in zip(a, b):
        np.testing.assert_array_equal(u, v)
    g = handle.loglik_batch(X, y, 2, P, 1.0)          # back on the Gaussian family the values differ
    assert not np.allclose(g[0], a[0])
