// Quadrature nodes of likeli.hyperpars (HX:554-556, ADV:557-559), callable from host AND device code.
//
// The reference takes them from two CRAN packages whose sources are not in its tree:
//   fOptions::runif.halton(N, 1)  -> base-2 radical inverse of 1..N
//   pscl::qigamma(p, alpha, beta) -> 1 / qgamma(1 - p, shape = alpha, rate = beta)
// Both are restated here from their mathematical definitions; tests compare against
// scipy.stats (tests/test_special.py).  The host instances serve ccgp_halton_base2 / ccgp_qigamma
// (the R surface, tests); the device instances serve ccgp_grid_marginal, which builds the whole
// G x N table of draws in HBM from the G x 4 hyperparameter matrix (round 3: before, the host
// expanded it -- 1.6 ms per distinct shape for the quantiles plus 50 MB over PCIe for the
// Heat-Exchanger grid, more than the evaluation itself).
#pragma once

#include <hip/hip_runtime.h>

namespace ccgp {

// base-2 radical inverse of i (i >= 1): runif.halton(N, 1)[i]
__host__ __device__ inline double halton2(unsigned i) {
  double f = 0.5, r = 0.0;
  for (unsigned k = i; k; k >>= 1) {
    if (k & 1u) r += f;
    f *= 0.5;
  }
  return r;
}

namespace special_detail {

// regularised incomplete gamma: lower P by series (x < a+1), upper Q by Lentz continued
// fraction (x >= a+1).  Returns the requested tail computed WITHOUT subtracting from 1
// whenever the direct form applies, so small tails keep full relative accuracy.
__host__ __device__ inline double log_prefactor(double a, double x) { return a * log(x) - x - lgamma(a); }

__host__ __device__ inline double p_series(double a, double x) {
  double sum = 1.0 / a, term = sum;
  for (int n = 1; n < 2000; ++n) {
    term *= x / (a + n);
    sum += term;
    if (fabs(term) < fabs(sum) * 1e-17) break;
  }
  return sum * exp(log_prefactor(a, x));
}

__host__ __device__ inline double q_contfrac(double a, double x) {
  const double tiny = 1e-300;
  double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
  for (int i = 1; i < 2000; ++i) {
    double an = -i * (i - a);
    b += 2.0;
    d = an * d + b;
    if (fabs(d) < tiny) d = tiny;
    c = b + an / c;
    if (fabs(c) < tiny) c = tiny;
    d = 1.0 / d;
    double del = d * c;
    h *= del;
    if (fabs(del - 1.0) < 1e-16) break;
  }
  return exp(log_prefactor(a, x)) * h;
}

__host__ __device__ inline double gamma_p(double a, double x) {
  if (x <= 0.0) return 0.0;
  return x < a + 1.0 ? p_series(a, x) : 1.0 - q_contfrac(a, x);
}
__host__ __device__ inline double gamma_q(double a, double x) {
  if (x <= 0.0) return 1.0;
  return x < a + 1.0 ? 1.0 - p_series(a, x) : q_contfrac(a, x);
}

}  // namespace special_detail

// x such that P(shape, x) = p, i.e. qgamma(p, shape, rate = 1).
__host__ __device__ inline double qgamma_unit(double p, double a) {
  using namespace special_detail;
  if (!(p > 0.0)) return 0.0;
  if (!(p < 1.0)) return __builtin_inf();
  const double q = 1.0 - p;
  // starting value (Wilson-Hilferty for a > 1, small-shape expansion otherwise)
  double x;
  if (a > 1.0) {
    const double pp = p < 0.5 ? p : q;
    const double t = sqrt(-2.0 * log(pp));
    double z = (2.30753 + t * 0.27061) / (1.0 + t * (0.99229 + t * 0.04481)) - t;
    if (p < 0.5) z = -z;
    const double w = 1.0 - 1.0 / (9.0 * a) - z / (3.0 * sqrt(a));
    x = a * w * w * w;
    if (x < 1e-3) x = 1e-3;
  } else {
    const double t = 1.0 - a * (0.253 + a * 0.12);
    x = p < t ? pow(p / t, 1.0 / a) : 1.0 - log(1.0 - (p - t) / (1.0 - t));
  }
  const double lg = lgamma(a);
  // Halley's iteration converges cubically from these starting values: four or five steps reach rounding level, where
  // the correction then wanders at 1 - 2 ulp and may never pass the relative test below.  On the device that is not a
  // detail: a wave runs until its SLOWEST lane is done, so a few lanes idling at rounding level made every wave of
  // grid_qtab_kernel walk all 100 iterations (round 4: 1.17 -> 0.08 ms for the Heat-Exchanger grid's table).  A
  // correction that no longer shrinks is rounding noise: stop there.
  double prev_step = __builtin_inf();
  for (int it = 0; it < 100; ++it) {
    if (x <= 0.0) x = 1e-300;
    // residual in the smaller tail
    const double err = p <= 0.5 ? gamma_p(a, x) - p : q - gamma_q(a, x);
    const double dens = exp(-x + (a - 1.0) * log(x) - lg);
    if (dens == 0.0) break;
    const double u = err / dens;
    double corr = u * ((a - 1.0) / x - 1.0);
    if (corr > 1.0) corr = 1.0;
    double dx = u / (1.0 - 0.5 * corr);
    double xn = x - dx;
    if (xn <= 0.0) xn = 0.5 * x;
    const double step = fabs(xn - x);
    x = xn;
    if (step <= 4e-16 * x) break;
    if (it >= 3 && step >= prev_step && step <= 1e-12 * x) break;
    prev_step = step;
  }
  return x;
}

// 1 / qgamma(1 - p, alpha, rate beta) = beta / qgamma_unit(1 - p, alpha)
__host__ __device__ inline double qigamma(double p, double alpha, double beta) {
  return beta / qgamma_unit(1.0 - p, alpha);
}

}  // namespace ccgp
