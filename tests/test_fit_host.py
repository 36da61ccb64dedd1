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
