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


def test_matern_quadrature_rule_of_the_device_kernel():
    """ccgp_internal.h matern_corr: z^nu K_nu(z) / (Gamma(nu) 2^(nu-1)) by the trapezoidal rule on
    int_0^inf exp(-z cosh t) cosh(nu t) dt with step 0.15 / max(1, sqrt z).  The same rule in numpy,
    against scipy's besselK (the oracle's restatement of base R's besselK, D1:350)."""
    import math
    from scipy import special as sps

    def rule(nu, z):
        if z * z <= 1e-12:
            return 1.0 - z * z / (4.0 * (nu - 1.0))
        hs = 0.15 / max(1.0, math.sqrt(z))
        s, k = 0.5, 1
        while True:
            t = k * hs
            c1 = math.cosh(t) - 1.0
            g = 0.5 * (math.exp(nu * t - z * c1) + math.exp(-nu * t - z * c1))   # exponents formed first: no overflow
            s += g
            k += 1
            if (g < 1e-17 * s and nu * t < z * c1) or k > 6000:
                break
        return math.exp(nu * math.log(z) - z) * hs * s / (math.gamma(nu) * 2.0 ** (nu - 1.0))

    for nu in (1.5, 2.5, 5.0, 7.0, 10.0):
        for z in np.concatenate([np.logspace(-6, 2.3, 40), [0.3, 1.0, 7.7, 33.3]]):
            want = float(orc.matern_corr(nu, z / (2.0 * math.sqrt(nu)), 1.0))     # h = z theta / (2 sqrt nu)
            assert abs(rule(nu, float(z)) - want) <= 5e-14 * max(want, 1e-30) + 1e-300, (nu, z)
