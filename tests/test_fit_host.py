"""Host-side inference layer (ccgp_amd/fit.py): the pieces that need no GPU."""
import math

import numpy as np
import pytest

from ccgp_amd import fit
from oracle import ccgp_oracle as orc


def test_host_jacobian_and_priors_match_the_oracle_restatement():
    t3 = np.array([[0.2, 1.5, -0.4], [-0.3, 2.0, 0.9]])
    for i in range(2):
        assert fit.log_jacobian(t3)[i] == pytest.approx(orc.log_jacobian(t3[i]), rel=1e-14)
        for script, pars in (("HX", (7, 3, 3, 28)), ("ADV", (4, 1.5, 6, 10)), ("GV", None), ("ISO", None), ("BSQ", None)):
            assert fit.log_prior(t3, script, pars)[i] == pytest.approx(orc.log_prior(t3[i], script, pars), rel=1e-14)
    t4 = np.array([[0.2, 1.5, -0.4, 0.7]])
    assert fit.log_prior(t4, "ANI")[0] == pytest.approx(orc.log_prior(t4[0], "ANI"), rel=1e-14)
    assert fit.log_jacobian(t4)[0] == pytest.approx(orc.log_jacobian(t4[0]), rel=1e-14)
    d = fit.transformed_to_draws(t4)[0]
    assert d == pytest.approx([1 / (1 + math.exp(0.4)), math.exp(0.2), math.exp(1.5), math.exp(0.7)])


def test_laplace_recovers_a_gaussian():
    mu = np.array([0.5, -1.0, 2.0])
    A = np.array([[2.0, 0.3, 0.0], [0.3, 1.0, -0.2], [0.0, -0.2, 0.5]])
    P = np.linalg.inv(A)

    def logp(rows):
        dlt = np.atleast_2d(rows) - mu
        return -0.5 * np.einsum("ij,jk,ik->i", dlt, P, dlt)

    est = fit.laplace(logp, np.zeros(3))
    np.testing.assert_allclose(est["mode"], mu, atol=2e-4)
    np.testing.assert_allclose(est["var"], A, rtol=1e-4, atol=1e-5)


def test_spectrum0_and_geweke_on_ar1():
    rng = np.random.default_rng(1)
    n, phi = 40000, 0.5
    e = rng.normal(size=n)
    x = np.empty(n)
    x[0] = e[0]
    for i in range(1, n):
        x[i] = phi * x[i - 1] + e[i]
    assert fit._spectrum0_ar(x) == pytest.approx(1.0 / (1 - phi) ** 2, rel=0.1)
    zs = [fit.geweke_z(x[i * 2000:(i + 1) * 2000]) for i in range(20)]
    assert abs(np.mean(zs)) < 0.8 and 0.5 < np.std(zs) < 1.6       # ~ N(0, 1) for a stationary chain
    drift = x[:2000] + np.linspace(0, 6, 2000)
    assert abs(fit.geweke_z(drift)) > 3                              # a drifting chain is flagged


@pytest.mark.parametrize("m", [2, 5])
def test_speculative_metro_is_the_sequential_chain_bit_for_bit(m):
    """Metro(speculate=m) evaluates 2^m - 1 candidates per device round trip; the chain, the betas,
    the counters and the generator state it leaves behind must equal the sequential run exactly."""
    mu = np.array([0.4, -0.7, 1.1])
    A = np.array([[0.30, 0.05, 0.00], [0.05, 0.20, -0.04], [0.00, -0.04, 0.10]])
    P = np.linalg.inv(A)
    calls = []

    def logpost_fn(rows):
        rows = np.atleast_2d(rows)
        calls.append(rows.shape[0])
        dlt = rows - mu
        val = -0.5 * np.einsum("ij,jk,ik->i", dlt, P, dlt)
        val = np.where(rows[:, 2] > 1.9, np.nan, val)        # an NA region (HX:454-455): never accepted
        return val, rows.sum(axis=1)

    args = (None, np.zeros(3), 400, 60, 10, 0.5, None, 1.0, None)
    r_seq, r_spec = np.random.default_rng(11), np.random.default_rng(11)
    seq = fit.Metro(*args, rng=r_seq, logpost_fn=logpost_fn)
    n_seq_calls = len(calls)
    del calls[:]
    spec = fit.Metro(*args, rng=r_spec, logpost_fn=logpost_fn, speculate=m)
    np.testing.assert_array_equal(seq["sample"], spec["sample"])
    np.testing.assert_array_equal(seq["beta"], spec["beta"])
    assert (seq["accepted"], seq["proposals"], seq["geweke_p"]) == (spec["accepted"], spec["proposals"], spec["geweke_p"])
    assert r_seq.random() == r_spec.random()                  # same generator state afterwards
    assert calls.count(2 ** m - 1) == spec["device_batches"]   # every chain round trip carries the whole tree
    assert spec["device_batches"] <= -(-seq["proposals"] // m) + 1 < seq["device_batches"]
    assert n_seq_calls > len(calls)


def test_results_table_has_the_reference_layout(tmp_path):
    """fit.write_results_table writes compare.GP's table in the layout of the reference's recorded
    `Size 50 Results 1.txt` (GV:759-761): same header, comparator columns NA."""
    import os
    from conftest import DATA
    from ccgp_amd.tables import read_table
    ref_names, ref = read_table(os.path.join(DATA, "gv", "results_50_1.txt"))
    m = 4
    table = dict(y_hat=np.arange(m) + 0.5, quant=np.full(m, 0.5), LL=np.arange(m) - 1.0, UL=np.arange(m) + 2.0,
                 y_true=np.arange(m) + 0.25)
    path = os.path.join(tmp_path, "res.txt")
    names = fit.write_results_table(path, table, ref[:m, :9], ref_names[:9])
    assert names == ref_names
    got_names, got = read_table(path)
    assert got_names == ref_names and got.shape == (m, len(ref_names))
    np.testing.assert_array_equal(got[:, :9], ref[:m, :9])
    np.testing.assert_array_equal(got[:, 9], table["y_hat"])
    assert np.isnan(got[:, 13:19]).all()
    np.testing.assert_array_equal(got[:, 19], table["y_true"])
