"""bench.py's synthetic-input generators and flop accounting (CPU only)."""
import importlib.util
import os

import numpy as np

from conftest import ROOT

spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_maximin_lhs_is_a_seeded_latin_hypercube():
    X = bench.maximin_lhs(256, 5, seed=3, sweeps=200)
    assert X.shape == (256, 5) and X.min() > 0 and X.max() < 1
    for k in range(5):                                   # one point per stratum in every dimension
        assert sorted(np.floor(X[:, k] * 256).astype(int).tolist()) == list(range(256))
    np.testing.assert_array_equal(X, bench.maximin_lhs(256, 5, seed=3, sweeps=200))
    base = bench.maximin_lhs(256, 5, seed=3, sweeps=0)

    def min_dist(A):
        d = ((A[:, None, :] - A[None, :, :]) ** 2).sum(-1)
        np.fill_diagonal(d, np.inf)
        return d.min()
    assert min_dist(X) >= min_dist(base)                 # the swaps never reduce the minimum distance


def test_cfg4_parameter_rows():
    X, y, P, K = bench.cfg4_inputs(16, n=128)
    assert X.shape == (128, 5) and y.shape == (128,) and P.shape == (16, 3 + 15) and K == 3
    assert np.all(P[:, :3] >= 0.15) and np.allclose(P[:, :3].sum(axis=1), 0.45 + 0.55)
    assert np.all(P[:, 3:] >= 0.5 - 1e-12) and np.all(P[:, 3:] <= 50 + 1e-9) and np.all(P[:, 13:] >= 20 - 1e-9)


def test_update_flops_accounting():
    # sum over block columns of (tiles below the diagonal + half a diagonal tile) x 2*128^3*j ~ n^3/3
    n = 4096
    f = bench.update_kernel_flops(n)
    assert 0.90 * n ** 3 / 3 < f < 1.0 * n ** 3 / 3
    assert bench.update_kernel_flops(128) == 0.0 and bench.update_kernel_flops(256) == 2.0 * 128 ** 3 * 0.5


def test_cfg2_and_cfg3_and_cfg5_shapes():
    X, y, P, K, s2 = bench.cfg2_inputs()
    assert X.shape == (64, 4) and P.shape == (624000, 10) and K == 2 and s2 > 0
    assert np.all(P[:, 0] + P[:, 1] == 1.0) and np.all(P[:, 2:] > 0)
    X3, y3, P3, K3, s23 = bench.cfg3_inputs()
    assert X3.shape == (100, 2) and P3.shape == (60 * 1728, 6) and np.allclose(P3[:, 4], 5 * P3[:, 2])
    sets, P5 = bench.cfg5_inputs(S=10)
    assert len(sets) == 17 and P5.shape == (10, 20)
    assert sorted({s[0].shape[0] for s in sets}) == [50, 90] and {s[2].shape[0] for s in sets} == {150, 110}
