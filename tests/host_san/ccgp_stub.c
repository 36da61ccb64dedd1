/* CPU STUB of the single-device entry points of include/ccgp.h -- TEST INFRASTRUCTURE for the host sanitizer runs
 * (tests/test_host_sanitizers.py; SURVEY.md section 5: "build-side ASan on the host shim").  It lets csrc/multi.cpp
 * (host threads, sharding, gathering into the caller's buffers) and r/ccgp_shim.c (SEXP unpacking, NA mapping, list
 * construction, CCGP_DEVICES parsing) run under AddressSanitizer / ThreadSanitizer on a machine without a GPU.
 * Never linked into libccgp.so, never used by the product path.
 *
 * What it keeps from the real library: every input buffer is READ in full and every output buffer WRITTEN in full
 * with the documented sizes (so a wrongly sized or offset buffer on the host side is an ASan report), results are
 * a deterministic function of the evaluation's own row (so sharded == unsharded can be asserted bit for bit), a
 * parameter row whose first weight is negative "fails" (status = 1, NaN), and device number 13 refuses every
 * batched call with CCGP_EHIP (a failing shard). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ccgp.h"

struct ccgp_handle {
  int device;
  int family;
  char err[128];
};

static double touch(const double* p, size_t n) {
  double s = 0.0;
  for (size_t i = 0; i < n; ++i) s += p[i] * (double)((i % 7) + 1);
  return s;
}

const char* ccgp_version(void) { return "ccgp stub (host sanitizers)"; }

int ccgp_create(int device, ccgp_handle** out) {
  if (!out) return CCGP_EINVAL;
  *out = NULL;
  if (device < 0 || device >= 16) return CCGP_EHIP;
  ccgp_handle* h = (ccgp_handle*)calloc(1, sizeof *h);
  h->device = device;
  *out = h;
  return CCGP_OK;
}
int ccgp_destroy(ccgp_handle* h) { free(h); return CCGP_OK; }
const char* ccgp_last_error(const ccgp_handle* h) { return h ? h->err : "null handle"; }
int ccgp_set_kernel(ccgp_handle* h, int family, double nu) {
  if (!h) return CCGP_EINVAL;
  if (family != 0 && !(nu > 1.0 && nu <= 10.0)) { snprintf(h->err, sizeof h->err, "ccgp_set_kernel: bad nu"); return CCGP_EINVAL; }
  h->family = family;
  return CCGP_OK;
}

static int refuse(ccgp_handle* h, const char* who) {
  if (h->device != 13) return 0;
  snprintf(h->err, sizeof h->err, "%s: stub device 13 always fails", who);
  return 1;
}

static double eval_row(const double* params, int ld, int b, int P, double base) {
  double v = base;
  for (int j = 0; j < P; ++j) v += params[b + (size_t)j * ld] * (double)(j + 1);
  return v;
}

int ccgp_loglik_batch(ccgp_handle* h, const double* X, int n, int d, const double* y, int K, const double* params, int B,
                      double sigma2, int mean_mode, double tau2, double* out_loglik, double* out_beta, int* status) {
  if (!h) return CCGP_EINVAL;
  if (n < 1 || d < 1 || K < 1 || K > 8 || B < 0 || !X || !y || !params || !out_loglik) {
    snprintf(h->err, sizeof h->err, "ccgp_loglik_batch: bad argument");
    return CCGP_EINVAL;
  }
  if (refuse(h, "ccgp_loglik_batch")) return CCGP_EHIP;
  const int P = K + K * d;
  const double base = touch(X, (size_t)n * d) + touch(y, n) + sigma2 + mean_mode + tau2 + h->family;
  int bad = 0;
  for (int b = 0; b < B; ++b) {
    const int fail = params[b] < 0.0;
    out_loglik[b] = fail ? NAN : eval_row(params, B, b, P, base);
    if (out_beta) out_beta[b] = fail ? NAN : 0.5 * out_loglik[b];
    if (status) status[b] = fail;
    bad += fail;
  }
  return bad;
}

int ccgp_grid_marginal(ccgp_handle* h, const double* X, int n, int d, const double* y, double sigma2, const double* hyper,
                       int G, int N, double tau, int take_log, double aniso_lambda, double* out, int* out_argmax,
                       double* out_logs) {
  if (!h) return CCGP_EINVAL;
  if (n < 1 || d < 1 || G < 1 || N < 1 || !X || !y || !hyper || !out) return CCGP_EINVAL;
  if (refuse(h, "ccgp_grid_marginal")) return CCGP_EHIP;
  const double base = touch(X, (size_t)n * d) + touch(y, n) + sigma2 + tau + take_log + aniso_lambda;
  int best = -1;
  for (int g = 0; g < G; ++g) {
    if (!(hyper[g] > 0.0)) { snprintf(h->err, sizeof h->err, "ccgp_grid_marginal: hyperparameters must be positive"); return CCGP_EINVAL; }
    out[g] = sin(base + eval_row(hyper, G, g, 4, 0.0));
    if (out_logs)
      for (int j = 0; j < N; ++j) out_logs[(size_t)g * N + j] = out[g] - j;
    if (best < 0 || out[g] > out[best]) best = g;
  }
  if (out_argmax) *out_argmax = best;
  return 0;
}

int ccgp_predict_batch(ccgp_handle* h, const double* X, int n, int d, const double* y, int K, const double* params, int S,
                       const double* Xtest, int m, double sigma2, double* out_mean, double* out_var, double* out_beta,
                       int* status) {
  if (!h) return CCGP_EINVAL;
  if (n < 1 || d < 1 || K < 1 || K > 8 || S < 1 || m < 1 || !X || !y || !params || !Xtest || !out_mean || !out_var) return CCGP_EINVAL;
  if (refuse(h, "ccgp_predict_batch")) return CCGP_EHIP;
  const int P = K + K * d;
  const double base = touch(X, (size_t)n * d) + touch(y, n) + sigma2 + h->family;
  int bad = 0;
  for (int s = 0; s < S; ++s) {
    const int fail = params[s] < 0.0;
    const double v = eval_row(params, S, s, P, base);
    for (int t = 0; t < m; ++t) {
      double xt = 0.0;
      for (int k = 0; k < d; ++k) xt += Xtest[t + (size_t)k * m] * (k + 1);
      out_mean[s + (size_t)t * S] = fail ? NAN : v + xt;
      out_var[s + (size_t)t * S] = fail ? NAN : v * v + xt;
    }
    if (out_beta) out_beta[s] = fail ? NAN : v;
    if (status) status[s] = fail;
    bad += fail;
  }
  return bad;
}

/* ---- the remaining entry points the R shim binds (single device only) ---------------------------------------- */
static int corr_any(ccgp_handle* h, const double* A, int m, const double* X, int n, int d, const double* pr, int P, double* out) {
  if (!h || m < 1 || n < 1 || d < 1 || !X || !pr || !out) { if (h) snprintf(h->err, sizeof h->err, "ccgp_corr_*: bad argument"); return CCGP_EINVAL; }
  const double s = touch(pr, P) + touch(X, (size_t)n * d) + (A ? touch(A, (size_t)m * d) : 0.0);
  for (size_t e = 0; e < (size_t)m * n; ++e) out[e] = cos(s + (double)e);
  return CCGP_OK;
}
int ccgp_corr_matrix(ccgp_handle* h, const double* X, int n, int d, const double* theta, double* out_R) {
  return corr_any(h, NULL, n, X, n, d, theta, d, out_R);
}
int ccgp_corr_cross(ccgp_handle* h, const double* Xnew, int m, const double* X, int n, int d, const double* theta, double* out) {
  return corr_any(h, Xnew, m, X, n, d, theta, d, out);
}
int ccgp_mixed_corr_matrix(ccgp_handle* h, const double* X, int n, int d, int K, const double* params, double* out_R) {
  if (h && (K < 1 || K > 8)) { snprintf(h->err, sizeof h->err, "ccgp_corr_*: bad argument"); return CCGP_EINVAL; }
  return corr_any(h, NULL, n, X, n, d, params, K + K * d, out_R);
}
int ccgp_mixed_corr_cross(ccgp_handle* h, const double* Xnew, int m, const double* X, int n, int d, int K, const double* params,
                          double* out) {
  if (h && (K < 1 || K > 8)) { snprintf(h->err, sizeof h->err, "ccgp_corr_*: bad argument"); return CCGP_EINVAL; }
  return corr_any(h, Xnew, m, X, n, d, params, K + K * d, out);
}
int ccgp_beta_mle(ccgp_handle* h, const double* R_inv, const double* y, int n, double* out_beta) {
  if (!h || !R_inv || !y || !out_beta) return CCGP_EINVAL;
  *out_beta = touch(R_inv, (size_t)n * n) + touch(y, n);
  return CCGP_OK;
}
int ccgp_sigma2_mle(ccgp_handle* h, const double* R_inv, const double* y, int n, double beta, double* out_sigma2) {
  if (!h || !R_inv || !y || !out_sigma2) return CCGP_EINVAL;
  *out_sigma2 = touch(R_inv, (size_t)n * n) + touch(y, n) + beta;
  return CCGP_OK;
}
int ccgp_factors(ccgp_handle* h, const double* R_inv, double beta, const double* y, int n, double* out) {
  if (!h || !R_inv || !y || !out) return CCGP_EINVAL;
  const double s = touch(R_inv, (size_t)n * n) + touch(y, n) + beta;
  for (int i = 0; i < 2 * n + 1; ++i) out[i] = s + i;
  return CCGP_OK;
}
int ccgp_predict_from_factors(ccgp_handle* h, const double* r, int m, int n, double beta, const double* mean_factor,
                              const double* var_factor1, double var_factor2, const double* R_inv, double sigma2, double* out_mean,
                              double* out_var) {
  if (!h || !r || !mean_factor || !var_factor1 || !R_inv || !out_mean || !out_var) return CCGP_EINVAL;
  const double s = touch(r, (size_t)m * n) + touch(mean_factor, n) + touch(var_factor1, n) + touch(R_inv, (size_t)n * n) + beta +
                   var_factor2 + sigma2;
  for (int t = 0; t < m; ++t) { out_mean[t] = s + t; out_var[t] = s - t; }
  return CCGP_OK;
}
int ccgp_predict_post(ccgp_handle* h, const double* Xnew, int m, const double* X, int n, int d, int K, const double* params_row,
                      double beta, const double* mean_factor, const double* var_factor1, double var_factor2, const double* R_inv,
                      double sigma2, double* out_mean, double* out_var) {
  if (!h || !Xnew || !X || !params_row || !mean_factor || !var_factor1 || !R_inv || !out_mean || !out_var || K < 1 || K > 8) return CCGP_EINVAL;
  const double s = touch(Xnew, (size_t)m * d) + touch(X, (size_t)n * d) + touch(params_row, K + K * d) + touch(mean_factor, n) +
                   touch(var_factor1, n) + touch(R_inv, (size_t)n * n) + beta + var_factor2 + sigma2 + h->family;
  for (int t = 0; t < m; ++t) { out_mean[t] = s + t; out_var[t] = s - t; }
  return CCGP_OK;
}
int ccgp_logpost(ccgp_handle* h, const double* X, int n, int d, const double* y, double sigma2, int prior_id, const double* theta_t,
                 const double* prior_pars, double* out_val, double* out_beta, double* out_loglik, double* out_Rinv, int* status) {
  if (!h || !X || !y || !theta_t || !out_val) return CCGP_EINVAL;
  const int nt = prior_id == CCGP_PRIOR_ANI ? 4 : 3;
  const double s = touch(X, (size_t)n * d) + touch(y, n) + touch(theta_t, nt) + sigma2 +
                   (prior_id == CCGP_PRIOR_INVGAMMA ? touch(prior_pars, 4) : 0.0);
  const int fail = theta_t[0] < -700.0;
  *out_val = s;
  if (out_beta) *out_beta = s + 1;
  if (out_loglik) *out_loglik = s + 2;
  if (out_Rinv) for (size_t e = 0; e < (size_t)n * n; ++e) out_Rinv[e] = s + (double)e;
  if (status) *status = fail;
  return fail;
}
int ccgp_logpost_batch(ccgp_handle* h, const double* X, int n, int d, const double* y, double sigma2, int prior_id,
                       const double* theta_t, int B, const double* prior_pars, double* out_val, double* out_beta, double* out_loglik,
                       int* status) {
  if (!h || !X || !y || !theta_t || !out_val || B < 1) return CCGP_EINVAL;
  const int q = prior_id == CCGP_PRIOR_ANI ? 4 : 3;
  const double s = touch(X, (size_t)n * d) + touch(y, n) + sigma2 + (prior_id == CCGP_PRIOR_INVGAMMA ? touch(prior_pars, 4) : 0.0);
  int bad = 0;
  for (int b = 0; b < B; ++b) {
    double t = 0.0;
    for (int k = 0; k < q; ++k) t += theta_t[b + (size_t)k * B];
    const int fail = theta_t[b] < -700.0;
    out_val[b] = fail ? NAN : -(t * t) + 1e-9 * s;        /* a peaked "posterior": some proposals accepted, some not */
    if (out_beta) out_beta[b] = t;
    if (out_loglik) out_loglik[b] = t + 2;
    if (status) status[b] = fail;
    bad += fail;
  }
  return bad;
}
int ccgp_mixed_logdet_designs(ccgp_handle* h, const double* Xs, int n, int d, int B, int K, const double* params, double* out_logdet,
                              int* status) {
  if (!h || !Xs || !params || !out_logdet || B < 1) return CCGP_EINVAL;
  const double s = touch(params, K + K * d);
  for (int b = 0; b < B; ++b) {
    out_logdet[b] = s + touch(Xs + (size_t)b * n * d, (size_t)n * d);
    if (status) status[b] = 0;
  }
  return 0;
}
