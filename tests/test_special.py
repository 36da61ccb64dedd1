"""Host-side quadrature nodes of likeli.hyperpars (HX:554-556): Halton base 2 and the
inverse-gamma quantile, checked against scipy (no GPU needed: these are host functions)."""
import numpy as np
import scipy.stats as sst

from ccgp_amd import api
from oracle import ccgp_oracle as orc


def test_halton_first_terms_and_oracle():
    u = api.halton_base2(1728)
    assert u[:7].tolist() == [0.5, 0.25, 0.75, 0.125, 0.625, 0.375, 0.875]
    np.testing.assert_array_equal(u, orc.runif_halton(1728))
    assert len(set(u.tolist())) == 1728 and u.min() > 0 and u.max() < 1


def test_qigamma_matches_scipy_on_every_grid_shape():
    u = api.halton_base2(1728)
    for alpha in (3.0, 4.0, 5.0, 6.0, 7.0, 8.0, 9.0, 11.0):
        for beta in (1.0, 4.0, 28.0, 250.0):
            got = api.qigamma(u, alpha, beta)
            want = sst.invgamma.ppf(u, alpha, scale=beta)
            np.testing.assert_allclose(got, want, rtol=2e-13, atol=0)
            np.testing.assert_allclose(got, orc.qigamma(u, alpha, beta), rtol=2e-13)


def test_qigamma_tails_and_small_shape():
    # pscl::qigamma is 1/qgamma(1 - p, ...): the 1 - p is formed in double first (HX:555), so the
    # yardstick for tiny p is qgamma at that rounded argument, not invgamma.ppf(p).
    p = np.array([1e-12, 1e-6, 0.5, 1 - 1e-6, 1 - 1e-12])
    for alpha in (0.3, 1.0, 2.5, 50.0):
        want = 2.0 / sst.gamma.ppf(1.0 - p, alpha)
        np.testing.assert_allclose(api.qigamma(p, alpha, 2.0), want, rtol=1e-10)
