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

#include <math.h>
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

/* logpost -> list(val, beta, R.Inv, loglik) -- HX:441-466 and its per-script variants.  want_rinv = FALSE (the slim
 * frame of r/ccgp.R: the prediction phase goes through ccgp_R_prediction_table and never reads R.Inv): solve(R) is
 * neither formed nor shipped, R.Inv is the 1 x 1 placeholder 0 that Metro stores per accepted draw (HX:520) */
SEXP ccgp_R_logpost(SEXP X, SEXP theta_t, SEXP y, SEXP sigma2, SEXP prior_id, SEXP prior_pars, SEXP want_rinv) {
  int n = Rf_nrows(X), d = Rf_ncols(X), status = 0;
  const int want = Rf_asInteger(want_rinv) != 0;
  double val = NA_REAL, beta = NA_REAL, ll = NA_REAL;
  SEXP Rinv = PROTECT(want ? Rf_allocMatrix(REALSXP, n, n) : Rf_ScalarReal(0.0));
  const double* pp = Rf_isNull(prior_pars) ? NULL : REAL(prior_pars);
  int rc = ccgp_logpost(handle(), REAL(X), n, d, REAL(y), Rf_asReal(sigma2), Rf_asInteger(prior_id),
                        REAL(theta_t), pp, &val, &beta, &ll, want ? REAL(Rinv) : NULL, &status);
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

/* The next m iterations of Metro's loop (HX:505-535 and its per-script copies) in ONE device call.
 * An iteration draws u <- runif(1) and theta.candidate <- rmnorm(1, theta.old, sqrt(2) V) = theta.old + e with
 * e = rnorm(q) %*% chol(sqrt(2) V): neither depends on whether the previous proposal was accepted, so R pre-draws m pairs
 * (u_t, e_t) in the script's order and every state the chain can be in after t proposals is theta.old plus a subset sum of
 * e_1 .. e_t.  The 2^m - 1 candidates are evaluated as one batch (ccgp_logpost_batch: the values the one-at-a-time logpost
 * would return, bit for bit) and the accept / reject walk `l.cand$val - l.old$val > log(u)` (HX:510-512) only reads them.
 *   prior = c(prior id, the script's prior parameters if it passes any), theta_old q, state_old = c(l.old$val, l.old$beta),
 *   u m, E m x q (row t = e_t)
 *   -> list(accepted logical m, theta m x q (state AFTER proposal t), val m (its value), beta m (its beta),
 *           cand.val m (the candidate's value: NA where its factorisation failed -- the reference's `if (NA > ...)` would stop
 *           the script; here such a proposal is rejected), evaluated = 2^m - 1)
 * Geweke's test between the iterations stays R code (r/ccgp.R walks the returned path). */
SEXP ccgp_R_metro_steps(SEXP X, SEXP y, SEXP sigma2, SEXP prior, SEXP theta_old, SEXP state_old, SEXP u, SEXP E) {
  const int n = Rf_nrows(X), d = Rf_ncols(X), q = Rf_length(theta_old), m = Rf_length(u);
  if (m < 1 || m > 10 || q < 3 || q > 4 || Rf_nrows(E) != m || Rf_ncols(E) != q || Rf_length(state_old) != 2 || Rf_length(prior) < 1)
    Rf_error("ccgp_R_metro_steps: u has %d entries (1 .. 10), theta.old %d (3 or 4), E must be %d x %d", m, q, m, q);
  const int B = (1 << m) - 1;
  double* cand = (double*)R_alloc((size_t)B * q, sizeof(double));        /* B x q column-major */
  double* states = (double*)R_alloc((size_t)(1 << m) * q, sizeof(double)); /* level t: 2^t states, history index = bits */
  double* val = (double*)R_alloc(B, sizeof(double));
  double* bet = (double*)R_alloc(B, sizeof(double));
  int* st = (int*)R_alloc(B, sizeof(int));
  const double* e = REAL(E);
  for (int k = 0; k < q; ++k) states[k] = REAL(theta_old)[k];
  int off = 0;
  for (int t = 0; t < m; ++t) {
    const int ns = 1 << t;
    /* candidates of level t, then the states of level t + 1: index 2 i = rejected (state i), 2 i + 1 = accepted (candidate i);
     * filled from the back so that the states are expanded in place */
    for (int i = ns - 1; i >= 0; --i)
      for (int k = 0; k < q; ++k) {
        const double s0 = states[(size_t)i * q + k];
        const double c = s0 + e[t + (size_t)k * m];
        cand[(off + i) + (size_t)k * B] = c;
        states[(size_t)(2 * i) * q + k] = s0;
        states[(size_t)(2 * i + 1) * q + k] = c;
      }
    off += ns;
  }
  const double* pp = Rf_length(prior) > 1 ? REAL(prior) + 1 : NULL;
  const int rc = ccgp_logpost_batch(handle(), REAL(X), n, d, REAL(y), Rf_asReal(sigma2), (int)REAL(prior)[0], cand, B, pp,
                                    val, bet, NULL, st);
  warn_rc(rc);
  SEXP acc = PROTECT(Rf_allocVector(LGLSXP, m));
  SEXP th = PROTECT(Rf_allocMatrix(REALSXP, m, q));
  SEXP vv = PROTECT(Rf_allocVector(REALSXP, m));
  SEXP bb = PROTECT(Rf_allocVector(REALSXP, m));
  SEXP cv = PROTECT(Rf_allocVector(REALSXP, m));
  double lo = REAL(state_old)[0], bo = REAL(state_old)[1];
  double cur[4];
  for (int k = 0; k < q; ++k) cur[k] = REAL(theta_old)[k];
  int idx = 0;
  off = 0;
  for (int t = 0; t < m; ++t) {
    const int c = off + idx;
    const double vc = (rc < 0 || st[c] != 0 || ISNAN(val[c])) ? NA_REAL : val[c];
    const int a = !ISNAN(vc) && (vc - lo) > log(REAL(u)[t]);
    if (a) {
      lo = vc;
      bo = bet[c];
      for (int k = 0; k < q; ++k) cur[k] = cand[c + (size_t)k * B];
    }
    LOGICAL(acc)[t] = a;
    REAL(cv)[t] = vc;
    REAL(vv)[t] = lo;
    REAL(bb)[t] = bo;
    for (int k = 0; k < q; ++k) REAL(th)[t + (size_t)k * m] = cur[k];
    off += 1 << t;
    idx = 2 * idx + a;
  }
  SEXP out = PROTECT(Rf_allocVector(VECSXP, 6));
  SEXP names = PROTECT(Rf_allocVector(STRSXP, 6));
  const char* nm[6] = {"accepted", "theta", "val", "beta", "cand.val", "evaluated"};
  for (int i = 0; i < 6; ++i) SET_STRING_ELT(names, i, Rf_mkChar(nm[i]));
  SET_VECTOR_ELT(out, 0, acc);
  SET_VECTOR_ELT(out, 1, th);
  SET_VECTOR_ELT(out, 2, vv);
  SET_VECTOR_ELT(out, 3, bb);
  SET_VECTOR_ELT(out, 4, cv);
  SET_VECTOR_ELT(out, 5, Rf_ScalarInteger(B));
  Rf_setAttrib(out, R_NamesSymbol, names);
  UNPROTECT(7);
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

/* ---- the prediction phase of the unchanged scripts -------------------------------------------------------------
 * prediction() runs apply(pars.frame, 1, predict.post, x.new = ...) for every test site (HX:688, GV:622) and
 * compare.GP runs prediction() for every row of D.test (HX:719, GV:671): S x m calls of predict.post, each re-parsing
 * an (5 + 2n + n^2)-wide frame row.  The stubs below take the frame factors.frame() returned (HX:625-644) -- a data
 * frame (list of columns) or a numeric matrix, columns p, theta1, theta2[, lambda] first -- and produce the whole
 * S x m table in ONE ccgp_predict_batch call; everything that indexes a frame or a frame row lives here, in C that
 * the test-suite executes, not in r/ccgp.R.
 *
 * layout: how a draw's leading columns map to the C-ABI parameter row
 *   0  (p, theta1, theta2), isotropic in d dimensions              HX:408-415, GV, ISO, BSQ; ADV's training kernel ADV:414-421
 *   1  ADV's predict.post AS WRITTEN: second scale theta1 (1 + pars[3])     ADV:672 (literal path only)
 *   2  (p, theta1, theta2, lambda), anisotropic, d = 2              ANI:399-406
 *   3  1-D Matern(nu) pair                                          D1:575-584
 *   4  1-D Matern(nu) + cubic spline                                D1F:453-462                              */
enum { LAYOUT_ISO = 0, LAYOUT_ADV_WRITTEN = 1, LAYOUT_ANI = 2, LAYOUT_D1 = 3, LAYOUT_D1F = 4 };

static int layout_ok(int layout, int d) {
  if (layout < LAYOUT_ISO || layout > LAYOUT_D1F) return 0;
  if (layout == LAYOUT_ANI && d != 2) return 0;
  if ((layout == LAYOUT_D1 || layout == LAYOUT_D1F) && d != 1) return 0;
  return 1;
}

/* one draw -> (w_1, w_2, theta_1k.., theta_2k..) written with stride ld (a column-major B x P matrix has ld = B) */
static void pack_draw(int layout, int d, double p, double t1, double t2, double lam, double* row, R_xlen_t ld) {
  row[0] = p;
  row[ld] = 1.0 - p;
  if (layout == LAYOUT_ANI) {
    row[2 * ld] = t1;
    row[3 * ld] = t2;
    row[4 * ld] = (1.0 + lam) * t1;
    row[5 * ld] = (1.0 + lam) * t2;
    return;
  }
  const double second = layout == LAYOUT_ADV_WRITTEN ? t1 * (1.0 + t2) : t2;
  for (int k = 0; k < d; ++k) {
    row[(2 + k) * ld] = t1;
    row[(2 + d + k) * ld] = second;
  }
}

/* a frame: data.frame (VECSXP of equally long numeric columns) or numeric matrix */
static int frame_rows(SEXP f) {
  if (TYPEOF(f) == VECSXP) return Rf_length(f) > 0 ? Rf_length(VECTOR_ELT(f, 0)) : 0;
  return Rf_nrows(f);
}
/* the leading `need` columns are numeric and as long as the first one (a list that is not a data frame may hold anything) */
static int frame_ok(SEXP f, int need) {
  if (TYPEOF(f) == REALSXP || TYPEOF(f) == INTSXP) return Rf_ncols(f) >= need;
  if (TYPEOF(f) != VECSXP || Rf_length(f) < need) return 0;
  const int S = Rf_length(VECTOR_ELT(f, 0));
  for (int j = 0; j < need; ++j) {
    SEXP col = VECTOR_ELT(f, j);
    if ((TYPEOF(col) != REALSXP && TYPEOF(col) != INTSXP) || Rf_length(col) != S) return 0;
  }
  return 1;
}
static double frame_get(SEXP f, int s, int j, int S) {
  SEXP col = f;
  R_xlen_t at = (R_xlen_t)s + (R_xlen_t)j * S;
  if (TYPEOF(f) == VECSXP) { col = VECTOR_ELT(f, j); at = s; }
  if (TYPEOF(col) == REALSXP) return REAL(col)[at];
  if (TYPEOF(col) == INTSXP) return INTEGER(col)[at] == NA_INTEGER ? NA_REAL : (double)INTEGER(col)[at];
  return NA_REAL;
}

/* the Matern / spline families of the 1-D scripts for the duration of one batched call */
static int family_of(int layout) { return layout == LAYOUT_D1 ? 1 : layout == LAYOUT_D1F ? 2 : 0; }
static int enter_family(int layout, double nu) {
  const int fam = family_of(layout);
  if (!fam) return 0;
  int rc = ccgp_set_kernel(handle(), fam, nu);
  if (rc == 0 && multi()) rc = ccgp_multi_set_kernel(g_multi, fam, nu);
  return rc;
}
static void leave_family(int layout) {
  if (!family_of(layout)) return;
  ccgp_set_kernel(handle(), 0, 0.0);
  if (multi()) ccgp_multi_set_kernel(g_multi, 0, 0.0);
}

/* scratch for the S x P parameter matrix of a frame; the caller keeps it PROTECTed until it returns */
static SEXP alloc_params(SEXP frame, SEXP Dtrain) {
  const int S = frame_rows(frame), d = Rf_ncols(Dtrain);
  return Rf_allocMatrix(REALSXP, S > 0 ? S : 1, 2 + 2 * (d > 0 ? d : 1));
}

/* mean / var: S x m column-major, beta: S (may be NULL).  Returns the library's code (< 0: nothing usable). */
static int table_into(SEXP frame, SEXP Dtrain, SEXP Dtest, SEXP sigma2, SEXP ytrain, int layout, double nu,
                      SEXP params, double* mean, double* var, double* beta) {
  const int n = Rf_nrows(Dtrain), d = Rf_ncols(Dtrain), m = Rf_nrows(Dtest), S = frame_rows(frame);
  const int need = layout == LAYOUT_ANI ? 4 : 3;
  if (!layout_ok(layout, d) || layout == LAYOUT_ADV_WRITTEN || S < 1 || m < 1 || !frame_ok(frame, need) ||
      Rf_ncols(Dtest) != d || Rf_length(ytrain) != n) {
    Rf_warning("libccgp: prediction table: frame / design shapes do not fit layout %d", layout);
    return CCGP_EINVAL;
  }
  for (int s = 0; s < S; ++s)
    pack_draw(layout, d, frame_get(frame, s, 0, S), frame_get(frame, s, 1, S), frame_get(frame, s, 2, S),
              need == 4 ? frame_get(frame, s, 3, S) : 0.0, REAL(params) + s, S);
  int rc = enter_family(layout, nu);
  if (rc < 0) {
    warn_rc(rc);
  } else if (multi()) {   /* sharded over the posterior draws */
    rc = ccgp_multi_predict_batch(g_multi, REAL(Dtrain), n, d, REAL(ytrain), 2, REAL(params), S, REAL(Dtest), m,
                                  Rf_asReal(sigma2), mean, var, beta, NULL);
    warn_rc_multi(rc);
  } else {
    rc = ccgp_predict_batch(handle(), REAL(Dtrain), n, d, REAL(ytrain), 2, REAL(params), S, REAL(Dtest), m,
                            Rf_asReal(sigma2), mean, var, beta, NULL);
    warn_rc(rc);
  }
  leave_family(layout);
  if (rc >= 0) {   /* a draw whose factorisation failed: NA, as the reference's R.Inv <- NA propagates (HX:454-455) */
    for (R_xlen_t i = 0; i < (R_xlen_t)S * m; ++i)
      if (ISNAN(mean[i]) || ISNAN(var[i])) { mean[i] = NA_REAL; var[i] = NA_REAL; }
  }
  return rc;
}

/* the (draw x test site) tables of a whole frame -> list(mean S x m, var S x m, beta S) */
SEXP ccgp_R_prediction_table(SEXP frame, SEXP Dtrain, SEXP Dtest, SEXP sigma2, SEXP ytrain, SEXP layout, SEXP nu) {
  const int S = frame_rows(frame) > 0 ? frame_rows(frame) : 1, m = Rf_nrows(Dtest) > 0 ? Rf_nrows(Dtest) : 1;
  SEXP mean = PROTECT(Rf_allocMatrix(REALSXP, S, m));
  SEXP var = PROTECT(Rf_allocMatrix(REALSXP, S, m));
  SEXP beta = PROTECT(Rf_allocVector(REALSXP, S));
  SEXP params = PROTECT(alloc_params(frame, Dtrain));
  int rc = table_into(frame, Dtrain, Dtest, sigma2, ytrain, Rf_asInteger(layout), Rf_asReal(nu), params, REAL(mean),
                      REAL(var), REAL(beta));
  if (rc < 0) { fill_na(REAL(mean), (R_xlen_t)S * m); fill_na(REAL(var), (R_xlen_t)S * m); fill_na(REAL(beta), S); }
  SEXP out = PROTECT(Rf_allocVector(VECSXP, 3));
  SET_VECTOR_ELT(out, 0, mean);
  SET_VECTOR_ELT(out, 1, var);
  SET_VECTOR_ELT(out, 2, beta);
  UNPROTECT(5);
  return out;
}

/* compare.GP's table, computed once and kept on the C side while the script's own compare.GP walks over the rows of
 * D.test (apply_pb(D.test, 1, prediction, ...), HX:719): ccgp_R_table_cache fills it, ccgp_R_table_lookup hands
 * prediction() the 2 x S block apply(pars.frame, 1, predict.post, ...) would have produced for x.new (rows mean, var:
 * HX:688 transposes it), ccgp_R_table_clear drops it (on.exit of the compare.GP wrapper). */
static struct {
  double *mean, *var, *Dtest;
  int S, m, d, next;
} g_tab = {NULL, NULL, NULL, 0, 0, 0, 0};

static void table_drop(void) {
  free(g_tab.mean); free(g_tab.var); free(g_tab.Dtest);
  g_tab.mean = g_tab.var = g_tab.Dtest = NULL;
  g_tab.S = g_tab.m = g_tab.d = g_tab.next = 0;
}

SEXP ccgp_R_table_clear(void) {
  table_drop();
  return R_NilValue;
}

/* -> number of draws cached (0: nothing cached, prediction() then takes the literal path) */
SEXP ccgp_R_table_cache(SEXP frame, SEXP Dtrain, SEXP Dtest, SEXP sigma2, SEXP ytrain, SEXP layout, SEXP nu) {
  table_drop();
  const int S = frame_rows(frame), m = Rf_nrows(Dtest), d = Rf_ncols(Dtest);
  if (S < 1 || m < 1 || d < 1) return Rf_ScalarInteger(0);
  g_tab.mean = (double*)malloc(sizeof(double) * (size_t)S * m);
  g_tab.var = (double*)malloc(sizeof(double) * (size_t)S * m);
  g_tab.Dtest = (double*)malloc(sizeof(double) * (size_t)m * d);
  if (!g_tab.mean || !g_tab.var || !g_tab.Dtest) {
    table_drop();
    Rf_warning("libccgp: no memory for a %d x %d prediction table", S, m);
    return Rf_ScalarInteger(0);
  }
  SEXP params = PROTECT(alloc_params(frame, Dtrain));
  const int rc = table_into(frame, Dtrain, Dtest, sigma2, ytrain, Rf_asInteger(layout), Rf_asReal(nu), params,
                            g_tab.mean, g_tab.var, NULL);
  if (rc < 0) {
    table_drop();
  } else {
    memcpy(g_tab.Dtest, REAL(Dtest), sizeof(double) * (size_t)m * d);
    g_tab.S = S; g_tab.m = m; g_tab.d = d; g_tab.next = 0;
  }
  SEXP out = Rf_ScalarInteger(rc < 0 ? 0 : S);
  UNPROTECT(1);
  return out;
}

/* x.new (d numbers) must be a row of the cached D.test, compared exactly; the search starts behind the previous hit
 * (apply walks the rows in order) and wraps.  NULL: no table, another frame size, or an unknown site. */
SEXP ccgp_R_table_lookup(SEXP xnew, SEXP S_) {
  if (!g_tab.mean || Rf_length(xnew) != g_tab.d || Rf_asInteger(S_) != g_tab.S) return R_NilValue;
  const double* x = REAL(xnew);
  int hit = -1;
  for (int q = 0; q < g_tab.m && hit < 0; ++q) {
    const int t = (g_tab.next + q) % g_tab.m;
    int same = 1;
    for (int k = 0; k < g_tab.d && same; ++k) same = g_tab.Dtest[t + (size_t)k * g_tab.m] == x[k];
    if (same) hit = t;
  }
  if (hit < 0) return R_NilValue;
  g_tab.next = (hit + 1) % g_tab.m;
  SEXP out = PROTECT(Rf_allocMatrix(REALSXP, 2, g_tab.S));
  for (int s = 0; s < g_tab.S; ++s) {
    REAL(out)[2 * (size_t)s] = g_tab.mean[s + (size_t)hit * g_tab.S];
    REAL(out)[2 * (size_t)s + 1] = g_tab.var[s + (size_t)hit * g_tab.S];
  }
  UNPROTECT(1);
  return out;
}

/* predict.post(x.new, D.train, pars, sigma2[, nu]) -- HX:655-673, ANI:604-623, ADV:660-678, D1:794-812, D1F:737-754:
 * pars is ONE frame row (p, theta1, theta2[, lambda], beta, mean.factor[n], var.factor1[n], var.factor2, R.Inv[n^2]);
 * one device round trip (ccgp_predict_post) -> cbind(mean, var), one row per site in x.new (a d-vector or an m x d
 * matrix).  A slim frame row (no cached terms) cannot be served here: warning + NA. */
SEXP ccgp_R_predict_post(SEXP xnew, SEXP Dtrain, SEXP pars, SEXP sigma2, SEXP layout_, SEXP nu) {
  const int n = Rf_nrows(Dtrain), d = Rf_ncols(Dtrain), layout = Rf_asInteger(layout_);
  const int has_dim = !Rf_isNull(Rf_getAttrib(xnew, R_DimSymbol));
  const int m = has_dim ? Rf_nrows(xnew) : 1;
  SEXP out = PROTECT(Rf_allocMatrix(REALSXP, m > 0 ? m : 1, 2)); /* cbind(mean, var) */
  const int o = layout == LAYOUT_ANI ? 4 : 3;   /* pars[o + 1] is beta (1-based), HX:659 / ANI:611 */
  const R_xlen_t want = (R_xlen_t)o + 2 + 2 * (R_xlen_t)n + (R_xlen_t)n * n;
  int rc = CCGP_EINVAL;
  if (!layout_ok(layout, d) || m < 1 || Rf_xlength(xnew) != (R_xlen_t)m * d) {
    Rf_warning("libccgp: predict.post: x.new / D.train do not fit layout %d", layout);
  } else if (Rf_xlength(pars) < want) {
    Rf_warning("libccgp: predict.post needs a full factors.frame row (%ld numbers for n = %d), got %ld: slim frames "
               "(ccgp.slim.frame) are served by prediction() / compare.GP / ccgp.prediction.table",
               (long)want, n, (long)Rf_xlength(pars));
  } else {
    const double* q = REAL(pars);
    double row[2 + 2 * 64];
    if (d > 64) {
      Rf_warning("libccgp: predict.post: more than 64 input dimensions");
    } else {
      pack_draw(layout, d, q[0], q[1], q[2], o == 4 ? q[3] : 0.0, row, 1);
      rc = enter_family(layout, Rf_asReal(nu));
      if (rc >= 0)
        rc = ccgp_predict_post(handle(), REAL(xnew), m, REAL(Dtrain), n, d, 2, row, q[o], q + o + 1, q + o + 1 + n,
                               q[o + 1 + 2 * n], q + o + 2 + 2 * n, Rf_asReal(sigma2), REAL(out), REAL(out) + m);
      warn_rc(rc);
      leave_family(layout);
    }
  }
  if (rc < 0) fill_na(REAL(out), 2 * (R_xlen_t)(m > 0 ? m : 1));
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
    {"ccgp_R_logpost", (DL_FUNC)&ccgp_R_logpost, 7},
    {"ccgp_R_metro_steps", (DL_FUNC)&ccgp_R_metro_steps, 8},
    {"ccgp_R_loglik_batch", (DL_FUNC)&ccgp_R_loglik_batch, 7},
    {"ccgp_R_grid_marginal", (DL_FUNC)&ccgp_R_grid_marginal, 8},
    {"ccgp_R_predict_batch", (DL_FUNC)&ccgp_R_predict_batch, 6},
    {"ccgp_R_predict_from_factors", (DL_FUNC)&ccgp_R_predict_from_factors, 7},
    {"ccgp_R_prediction_table", (DL_FUNC)&ccgp_R_prediction_table, 7},
    {"ccgp_R_table_cache", (DL_FUNC)&ccgp_R_table_cache, 7},
    {"ccgp_R_table_lookup", (DL_FUNC)&ccgp_R_table_lookup, 2},
    {"ccgp_R_table_clear", (DL_FUNC)&ccgp_R_table_clear, 0},
    {"ccgp_R_predict_post", (DL_FUNC)&ccgp_R_predict_post, 6},
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
  table_drop();
}
