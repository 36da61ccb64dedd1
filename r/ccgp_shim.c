/*
 * .Call() shim between R and libccgp (include/ccgp.h).
 *
 * NOT compiled against R in this repository's build: the build container has no R (no
 * Rinternals.h, no libR.so).  The test-suite compiles it against include/ccgp.h and a FUNCTIONAL mock of the
 * R API (tests/r_mock/: typed vectors, dim / names, NA_REAL, PROTECT-stack and GC-hazard accounting, captured
 * warnings, registered-routine dispatch) and executes every routine below on the GPU through that mock
 * (tests/test_gpu_r_shim.py); under AddressSanitizer / ThreadSanitizer against a stub device library on the CPU
 * (tests/test_host_sanitizers.py).  A maintainer builds it on a machine with R and ROCm:
 *
 *     R CMD SHLIB -o ccgpR.so r/ccgp_shim.c -I include \
 *         -L convex-combination-of-gaussian-processes_amd/csrc -lccgp
 *
 * and loads it from r/ccgp.R (dyn.load).  R matrices are already column-major fp64, so
 * REAL() pointers go straight through; results are PROTECTed allocVector()s.  Errors
 * never longjmp out of a device call: a negative return code becomes Rf_warning() plus
 * NA_real_ results, mirroring the reference's try(solve(R)) -> NA convention
 * (Heat Exchanger Emulator/Combined GP Heat Exchanger.R:454-455).
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>

#include <stdlib.h>
#include <string.h>

#include "ccgp.h"

static ccgp_handle* g_handle = NULL; /* R is single-threaded: one handle per process */
static ccgp_multi* g_multi = NULL;   /* CCGP_DEVICES=k (k > 1) or a comma list: the batched calls are sharded */
static int g_multi_tried = 0;

static ccgp_handle* handle(void) {
  if (!g_handle) {
    int dev = 0;
    const char* e = getenv("CCGP_DEVICE");
    if (e) dev = atoi(e);
    if (ccgp_create(dev, &g_handle) != CCGP_OK) {
      g_handle = NULL;
      Rf_error("libccgp: no HIP device %d (there is no CPU fallback)", dev);
    }
  }
  return g_handle;
}

/* CCGP_DEVICES="8" -> devices 0..7;  CCGP_DEVICES="0,2,4" -> that list.  NULL when unset or a single device:
 * the batched entry points then use the one handle. */
static ccgp_multi* multi(void) {
  if (!g_multi_tried) {
    g_multi_tried = 1;
    const char* e = getenv("CCGP_DEVICES");
    if (e && *e) {
      int devs[64], k = 0;
      if (strchr(e, ',')) {
        const char* q = e;
        while (*q && k < 64) {
          devs[k++] = atoi(q);
          q = strchr(q, ',');
          if (!q) break;
          ++q;
        }
      } else {
        k = atoi(e);
        if (k > 64) k = 64;
        for (int i = 0; i < k; ++i) devs[i] = i;
      }
      if (k > 1 && ccgp_multi_create(k, devs, &g_multi) != CCGP_OK) {
        g_multi = NULL;
        Rf_warning("libccgp: CCGP_DEVICES=%s could not be opened; using one device", e);
      }
    }
  }
  return g_multi;
}

static void warn_rc(int rc) {
  if (rc < 0) Rf_warning("libccgp error %d: %s", rc, ccgp_last_error(g_handle));
}
static void warn_rc_multi(int rc) {
  if (rc < 0) Rf_warning("libccgp error %d: %s", rc, ccgp_multi_last_error(g_multi));
}

static void fill_na(double* p, R_xlen_t n) {
  for (R_xlen_t i = 0; i < n; ++i) p[i] = NA_REAL;
}

/* corr.matrix(X, theta) -- HX:328-337 / ANI:351-360 (theta has ncol(X) entries) */
SEXP ccgp_R_corr_matrix(SEXP X, SEXP theta) {
  int n = Rf_nrows(X), d = Rf_ncols(X);
  SEXP out = PROTECT(Rf_allocMatrix(REALSXP, n, n));
  int rc = ccgp_corr_matrix(handle(), REAL(X), n, d, REAL(theta), REAL(out));
  if (rc < 0) { warn_rc(rc); fill_na(REAL(out), (R_xlen_t)n * n); }
  UNPROTECT(1);
  return out;
}

/* corr.vec(x, X, theta) for one or many new sites (rows of Xnew) -- HX:367-375 */
SEXP ccgp_R_corr_cross(SEXP Xnew, SEXP X, SEXP theta) {
  int m = Rf_nrows(Xnew), n = Rf_nrows(X), d = Rf_ncols(X);
  SEXP out = PROTECT(Rf_allocMatrix(REALSXP, m, n));
  int rc = ccgp_corr_cross(handle(), REAL(Xnew), m, REAL(X), n, d, REAL(theta), REAL(out));
  if (rc < 0) { warn_rc(rc); fill_na(REAL(out), (R_xlen_t)m * n); }
  UNPROTECT(1);
  return out;
}

/* Mixed.corr.matrix -- HX:408-415; params = c(w, theta rows) as in include/ccgp.h */
SEXP ccgp_R_mixed_corr_matrix(SEXP X, SEXP K, SEXP params) {
  int n = Rf_nrows(X), d = Rf_ncols(X);
  SEXP out = PROTECT(Rf_allocMatrix(REALSXP, n, n));
  int rc = ccgp_mixed_corr_matrix(handle(), REAL(X), n, d, Rf_asInteger(K), REAL(params), REAL(out));
  if (rc < 0) { warn_rc(rc); fill_na(REAL(out), (R_xlen_t)n * n); }
  UNPROTECT(1);
  return out;
}

/* Mixed.corr.vec -- HX:425-431 */
SEXP ccgp_R_mixed_corr_cross(SEXP Xnew, SEXP X, SEXP K, SEXP params) {
  int m = Rf_nrows(Xnew), n = Rf_nrows(X), d = Rf_ncols(X);
  SEXP out = PROTECT(Rf_allocMatrix(REALSXP, m, n));
  int rc = ccgp_mixed_corr_cross(handle(), REAL(Xnew), m, REAL(X), n, d, Rf_asInteger(K), REAL(params),
                                 REAL(out));
  if (rc < 0) { warn_rc(rc); fill_na(REAL(out), (R_xlen_t)m * n); }
  UNPROTECT(1);
  return out;
}

/* logpost -> list(val, beta, R.Inv, loglik) -- HX:441-466 and its per-script variants */
SEXP ccgp_R_logpost(SEXP X, SEXP theta_t, SEXP y, SEXP sigma2, SEXP prior_id, SEXP prior_pars) {
  int n = Rf_nrows(X), d = Rf_ncols(X), status = 0;
  double val = NA_REAL, beta = NA_REAL, ll = NA_REAL;
  SEXP Rinv = PROTECT(Rf_allocMatrix(REALSXP, n, n));
  const double* pp = Rf_isNull(prior_pars) ? NULL : REAL(prior_pars);
  int rc = ccgp_logpost(handle(), REAL(X), n, d, REAL(y), Rf_asReal(sigma2), Rf_asInteger(prior_id),
                        REAL(theta_t), pp, &val, &beta, &ll, REAL(Rinv), &status);
  warn_rc(rc);
  SEXP out = PROTECT(Rf_allocVector(VECSXP, 4));
  SEXP names = PROTECT(Rf_allocVector(STRSXP, 4));
  SET_STRING_ELT(names, 0, Rf_mkChar("val"));
  SET_STRING_ELT(names, 1, Rf_mkChar("beta"));
  SET_STRING_ELT(names, 2, Rf_mkChar("R.Inv"));
  SET_STRING_ELT(names, 3, Rf_mkChar("loglik"));
  if (rc != 0 || status != 0) { val = NA_REAL; beta = NA_REAL; ll = NA_REAL; }
  SET_VECTOR_ELT(out, 0, Rf_ScalarReal(val));
  SET_VECTOR_ELT(out, 1, Rf_ScalarReal(beta));
  SET_VECTOR_ELT(out, 2, (rc != 0 || status != 0) ? Rf_ScalarLogical(NA_LOGICAL) : Rinv); /* R.Inv <- NA */
  SET_VECTOR_ELT(out, 3, Rf_ScalarReal(ll));
  Rf_setAttrib(out, R_NamesSymbol, names);
  UNPROTECT(3);
  return out;
}

/* batched likelihood: params is B x P -> list(loglik, beta, status) */
SEXP ccgp_R_loglik_batch(SEXP X, SEXP y, SEXP K, SEXP params, SEXP sigma2, SEXP mean_mode, SEXP tau2) {
  int n = Rf_nrows(X), d = Rf_ncols(X), B = Rf_nrows(params);
  SEXP ll = PROTECT(Rf_allocVector(REALSXP, B));
  SEXP beta = PROTECT(Rf_allocVector(REALSXP, B));
  SEXP st = PROTECT(Rf_allocVector(INTSXP, B));
  int rc;
  if (multi()) {
    rc = ccgp_multi_loglik_batch(g_multi, REAL(X), n, d, REAL(y), Rf_asInteger(K), REAL(params), B,
                                 Rf_asReal(sigma2), Rf_asInteger(mean_mode), Rf_asReal(tau2), REAL(ll),
                                 REAL(beta), INTEGER(st));
    warn_rc_multi(rc);
  } else {
    rc = ccgp_loglik_batch(handle(), REAL(X), n, d, REAL(y), Rf_asInteger(K), REAL(params), B,
                           Rf_asReal(sigma2), Rf_asInteger(mean_mode), Rf_asReal(tau2), REAL(ll),
                           REAL(beta), INTEGER(st));
    warn_rc(rc);
  }
  if (rc < 0) { fill_na(REAL(ll), B); fill_na(REAL(beta), B); }
  for (int i = 0; i < B; ++i)
    if (ISNAN(REAL(ll)[i])) { REAL(ll)[i] = NA_REAL; REAL(beta)[i] = NA_REAL; }
  SEXP out = PROTECT(Rf_allocVector(VECSXP, 3));
  SET_VECTOR_ELT(out, 0, ll);
  SET_VECTOR_ELT(out, 1, beta);
  SET_VECTOR_ELT(out, 2, st);
  UNPROTECT(4);
  return out;
}

/* choose.hyperpars / likeli.hyperpars -- HX:549-595, ADV:552-599 */
SEXP ccgp_R_grid_marginal(SEXP X, SEXP y, SEXP sigma2, SEXP hyper, SEXP N, SEXP tau, SEXP take_log,
                          SEXP aniso_lambda) {
  int n = Rf_nrows(X), d = Rf_ncols(X), G = Rf_nrows(hyper), arg = -1;
  SEXP vals = PROTECT(Rf_allocVector(REALSXP, G));
  int rc;
  if (multi()) {   /* sharded by grid row over the devices of CCGP_DEVICES */
    rc = ccgp_multi_grid_marginal(g_multi, REAL(X), n, d, REAL(y), Rf_asReal(sigma2), REAL(hyper), G,
                                  Rf_asInteger(N), Rf_asReal(tau), Rf_asInteger(take_log),
                                  Rf_asReal(aniso_lambda), REAL(vals), &arg, NULL);
    warn_rc_multi(rc);
  } else {
    rc = ccgp_grid_marginal(handle(), REAL(X), n, d, REAL(y), Rf_asReal(sigma2), REAL(hyper), G,
                            Rf_asInteger(N), Rf_asReal(tau), Rf_asInteger(take_log),
                            Rf_asReal(aniso_lambda), REAL(vals), &arg, NULL);
    warn_rc(rc);
  }
  if (rc < 0) fill_na(REAL(vals), G);
  SEXP out = PROTECT(Rf_allocVector(VECSXP, 2));
  SET_VECTOR_ELT(out, 0, vals);
  SET_VECTOR_ELT(out, 1, Rf_ScalarInteger(arg + 1)); /* which.max is 1-based in R */
  UNPROTECT(2);
  return out;
}

/* (draw x test point) predictive mean / variance tables -- HX:655-693 */
SEXP ccgp_R_predict_batch(SEXP X, SEXP y, SEXP K, SEXP params, SEXP Xtest, SEXP sigma2) {
  int n = Rf_nrows(X), d = Rf_ncols(X), S = Rf_nrows(params), m = Rf_nrows(Xtest);
  SEXP mean = PROTECT(Rf_allocMatrix(REALSXP, S, m));
  SEXP var = PROTECT(Rf_allocMatrix(REALSXP, S, m));
  SEXP beta = PROTECT(Rf_allocVector(REALSXP, S));
  int rc;
  if (multi()) {   /* sharded over the posterior draws */
    rc = ccgp_multi_predict_batch(g_multi, REAL(X), n, d, REAL(y), Rf_asInteger(K), REAL(params), S,
                                  REAL(Xtest), m, Rf_asReal(sigma2), REAL(mean), REAL(var), REAL(beta), NULL);
    warn_rc_multi(rc);
  } else {
    rc = ccgp_predict_batch(handle(), REAL(X), n, d, REAL(y), Rf_asInteger(K), REAL(params), S,
                            REAL(Xtest), m, Rf_asReal(sigma2), REAL(mean), REAL(var), REAL(beta), NULL);
    warn_rc(rc);
  }
  if (rc < 0) { fill_na(REAL(mean), (R_xlen_t)S * m); fill_na(REAL(var), (R_xlen_t)S * m); }
  SEXP out = PROTECT(Rf_allocVector(VECSXP, 3));
  SET_VECTOR_ELT(out, 0, mean);
  SET_VECTOR_ELT(out, 1, var);
  SET_VECTOR_ELT(out, 2, beta);
  UNPROTECT(4);
  return out;
}

/* literal predict.post arithmetic with the caller's cached R.Inv terms -- HX:667-670 */
SEXP ccgp_R_predict_from_factors(SEXP r, SEXP beta, SEXP mean_factor, SEXP var_factor1, SEXP var_factor2,
                                 SEXP R_inv, SEXP sigma2) {
  int m = Rf_nrows(r), n = Rf_ncols(r);
  SEXP out = PROTECT(Rf_allocMatrix(REALSXP, m, 2)); /* cbind(mean, var) */
  int rc = ccgp_predict_from_factors(handle(), REAL(r), m, n, Rf_asReal(beta), REAL(mean_factor),
                                     REAL(var_factor1), Rf_asReal(var_factor2), REAL(R_inv),
                                     Rf_asReal(sigma2), REAL(out), REAL(out) + m);
  if (rc < 0) { warn_rc(rc); fill_na(REAL(out), (R_xlen_t)2 * m); }
  UNPROTECT(1);
  return out;
}

/* factors(MCMC.data, n.train, y.train) -- HX:604-613 */
SEXP ccgp_R_factors(SEXP R_inv, SEXP beta, SEXP y) {
  int n = Rf_length(y);
  SEXP out = PROTECT(Rf_allocVector(REALSXP, 2 * n + 1));
  int rc = ccgp_factors(handle(), REAL(R_inv), Rf_asReal(beta), REAL(y), n, REAL(out));
  if (rc < 0) { warn_rc(rc); fill_na(REAL(out), 2 * n + 1); }
  UNPROTECT(1);
  return out;
}

/* beta.MLE(R.Inv, y) -- HX:384-388 ; sigma2.MLE(R.Inv, y, beta) -- HX:394-399 */
SEXP ccgp_R_beta_mle(SEXP R_inv, SEXP y) {
  double b = NA_REAL;
  warn_rc(ccgp_beta_mle(handle(), REAL(R_inv), REAL(y), Rf_length(y), &b));
  return Rf_ScalarReal(b);
}
SEXP ccgp_R_sigma2_mle(SEXP R_inv, SEXP y, SEXP beta) {
  double s = NA_REAL;
  warn_rc(ccgp_sigma2_mle(handle(), REAL(R_inv), REAL(y), Rf_length(y), Rf_asReal(beta), &s));
  return Rf_ScalarReal(s);
}

/* log det R.mixed for B candidate designs (a list of n x d matrices flattened into an n*d x B matrix)
 * -- Entropy / Augmented.Mixed.Entropy, Batch Sequential ME Design.R:856-877 */
SEXP ccgp_R_mixed_logdet_designs(SEXP Xs, SEXP n, SEXP d, SEXP K, SEXP params) {
  int B = Rf_ncols(Xs);
  SEXP out = PROTECT(Rf_allocVector(REALSXP, B));
  int rc = ccgp_mixed_logdet_designs(handle(), REAL(Xs), Rf_asInteger(n), Rf_asInteger(d), B,
                                     Rf_asInteger(K), REAL(params), REAL(out), NULL);
  if (rc < 0) { warn_rc(rc); fill_na(REAL(out), B); }
  for (int i = 0; i < B; ++i) if (ISNAN(REAL(out)[i])) REAL(out)[i] = NA_REAL;
  UNPROTECT(1);
  return out;
}

/* correlation family for the calls that follow: 0 = Gaussian, 1 = Matern(nu) of the 1-D script
 * (Matern.corr.func, 1D Combined GP Public.R:348-351), 2 = Matern(nu) + cubic spline of the two-family script
 * (1D Combined GP Two Families Public.R:346-357, 453-462) */
SEXP ccgp_R_set_kernel(SEXP family, SEXP nu) {
  int rc = ccgp_set_kernel(handle(), Rf_asInteger(family), Rf_asReal(nu));
  warn_rc(rc);
  if (rc == 0 && multi()) {
    rc = ccgp_multi_set_kernel(g_multi, Rf_asInteger(family), Rf_asReal(nu));
    warn_rc_multi(rc);
  }
  return Rf_ScalarInteger(rc);
}

/* how many devices the batched calls are sharded over (1 = the single handle) */
SEXP ccgp_R_devices(void) {
  ccgp_multi* m = multi();
  return Rf_ScalarInteger(m ? ccgp_multi_count(m) : 1);
}

static const R_CallMethodDef call_methods[] = {
    {"ccgp_R_corr_matrix", (DL_FUNC)&ccgp_R_corr_matrix, 2},
    {"ccgp_R_corr_cross", (DL_FUNC)&ccgp_R_corr_cross, 3},
    {"ccgp_R_mixed_corr_matrix", (DL_FUNC)&ccgp_R_mixed_corr_matrix, 3},
    {"ccgp_R_mixed_corr_cross", (DL_FUNC)&ccgp_R_mixed_corr_cross, 4},
    {"ccgp_R_logpost", (DL_FUNC)&ccgp_R_logpost, 6},
    {"ccgp_R_loglik_batch", (DL_FUNC)&ccgp_R_loglik_batch, 7},
    {"ccgp_R_grid_marginal", (DL_FUNC)&ccgp_R_grid_marginal, 8},
    {"ccgp_R_predict_batch", (DL_FUNC)&ccgp_R_predict_batch, 6},
    {"ccgp_R_predict_from_factors", (DL_FUNC)&ccgp_R_predict_from_factors, 7},
    {"ccgp_R_factors", (DL_FUNC)&ccgp_R_factors, 3},
    {"ccgp_R_beta_mle", (DL_FUNC)&ccgp_R_beta_mle, 2},
    {"ccgp_R_sigma2_mle", (DL_FUNC)&ccgp_R_sigma2_mle, 3},
    {"ccgp_R_mixed_logdet_designs", (DL_FUNC)&ccgp_R_mixed_logdet_designs, 5},
    {"ccgp_R_set_kernel", (DL_FUNC)&ccgp_R_set_kernel, 2},
    {"ccgp_R_devices", (DL_FUNC)&ccgp_R_devices, 0},
    {NULL, NULL, 0}};

void R_init_ccgpR(DllInfo* dll) {
  R_registerRoutines(dll, NULL, call_methods, NULL, NULL);
  R_useDynamicSymbols(dll, FALSE);
}

void R_unload_ccgpR(DllInfo* dll) {
  (void)dll;
  if (g_handle) { ccgp_destroy(g_handle); g_handle = NULL; }
  if (g_multi) { ccgp_multi_destroy(g_multi); g_multi = NULL; }
  g_multi_tried = 0;   /* CCGP_DEVICES is read again after a reload */
}
