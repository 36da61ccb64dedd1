/* Driver of the host sanitizer runs (tests/test_host_sanitizers.py): csrc/multi.cpp and r/ccgp_shim.c (against the
 * functional R-API mock of tests/r_mock/) on top of the CPU stub of the device entry points (ccgp_stub.c), built with
 * -fsanitize=address,undefined or -fsanitize=thread.  Every scenario checks results, not only "no report":
 * sharded == unsharded bit for bit (1 / 2 / 3 / 8 shards, ragged, fewer items than shards, a failing evaluation), a
 * failing shard surfaces as its negative code with the shard named, and the shim turns that into a warning + NA.
 * Prints "host-sanitizer driver: OK" and exits 0 when everything held. */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <R.h>
#include <Rinternals.h>

#include "ccgp.h"

/* harness entry points of rmock.c */
int rmock_load(void);
void rmock_unload(void);
SEXP rmock_dot_call(const char* name, int nargs, SEXP* a);
SEXP rmock_real(const double* v, R_xlen_t n, int nrow, int ncol);
SEXP rmock_int(const int* v, R_xlen_t n);
SEXP rmock_nil(void);
SEXP rmock_list(R_xlen_t n);
void rmock_list_set(SEXP l, R_xlen_t i, SEXP v);
void rmock_list_names(SEXP l, const char** names, int n_names);
int rmock_typeof(SEXP x);
SEXP rmock_elt(SEXP x, long i);
void* rmock_data(SEXP x);
long rmock_length(SEXP x);
int rmock_is_na_real(double x);
int rmock_n_warnings(void);
const char* rmock_warning(int i);
const char* rmock_last_error(void);
int rmock_counter(int which);
void rmock_reset(void);

static int failures = 0;
#define CHECK(cond, ...)                                                     \
  do {                                                                       \
    if (!(cond)) {                                                           \
      ++failures;                                                            \
      fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__);                 \
      fprintf(stderr, __VA_ARGS__);                                          \
      fprintf(stderr, "\n");                                                 \
    }                                                                        \
  } while (0)

static double* filled(size_t n, double seed) {
  double* p = (double*)malloc(sizeof(double) * (n ? n : 1));
  for (size_t i = 0; i < n; ++i) p[i] = seed + 0.37 * (double)i - floor(0.11 * (double)i);
  return p;
}
static int same(const double* a, const double* b, size_t n) {   /* bitwise, NaN == NaN */
  return memcmp(a, b, sizeof(double) * n) == 0;
}

enum { N = 9, D = 3, K = 2, P = K + K * D };

static void multi_scenarios(void) {
  double* X = filled((size_t)N * D, 0.1);
  double* y = filled(N, 2.0);
  ccgp_handle* one = NULL;
  CHECK(ccgp_create(0, &one) == CCGP_OK, "create");
  const int shard_counts[] = {1, 2, 3, 8};
  const int batch_sizes[] = {13, 2, 8, 1};
  for (int si = 0; si < 4; ++si) {
    const int w = shard_counts[si];
    ccgp_multi* m = NULL;
    CHECK(ccgp_multi_create(w, NULL, &m) == CCGP_OK && ccgp_multi_count(m) == w, "multi_create(%d)", w);
    for (int bi = 0; bi < 4; ++bi) {
      const int B = batch_sizes[bi];
      double* params = filled((size_t)B * P, 0.5);
      if (B > 4) params[3] = -1.0;                 /* evaluation 3 fails: status 1, NaN */
      double *ll0 = filled(B, 0), *b0 = filled(B, 0), *ll1 = filled(B, 1), *b1 = filled(B, 1);
      int* st0 = (int*)calloc(B, sizeof(int));
      int* st1 = (int*)calloc(B, sizeof(int));
      const int r0 = ccgp_loglik_batch(one, X, N, D, y, K, params, B, 1.5, 1, 4.0, ll0, b0, st0);
      const int r1 = ccgp_multi_loglik_batch(m, X, N, D, y, K, params, B, 1.5, 1, 4.0, ll1, b1, st1);
      CHECK(r0 == r1 && r0 == (B > 4), "loglik return %d vs %d (w=%d B=%d)", r0, r1, w, B);
      CHECK(same(ll0, ll1, B) && same(b0, b1, B) && memcmp(st0, st1, sizeof(int) * B) == 0, "loglik w=%d B=%d", w, B);
      /* optional outputs absent */
      CHECK(ccgp_multi_loglik_batch(m, X, N, D, y, K, params, B, 1.5, 1, 4.0, ll1, NULL, NULL) == r0 && same(ll0, ll1, B),
            "loglik without beta / status");
      free(params); free(ll0); free(b0); free(ll1); free(b1); free(st0); free(st1);
    }
    /* grid: G rows, argmax and the G x N log table */
    for (int G = 1; G <= 11; G += 5) {
      const int Nn = 4;
      double* hyper = filled((size_t)G * 4, 1.0);
      double *o0 = filled(G, 0), *o1 = filled(G, 1), *l0 = filled((size_t)G * Nn, 0), *l1 = filled((size_t)G * Nn, 1);
      int a0 = -7, a1 = -9;
      CHECK(ccgp_grid_marginal(one, X, N, D, y, 2.0, hyper, G, Nn, 50.0, 1, -1.0, o0, &a0, l0) == 0, "grid single");
      CHECK(ccgp_multi_grid_marginal(m, X, N, D, y, 2.0, hyper, G, Nn, 50.0, 1, -1.0, o1, &a1, l1) == 0, "grid multi");
      CHECK(same(o0, o1, G) && a0 == a1 && same(l0, l1, (size_t)G * Nn), "grid w=%d G=%d", w, G);
      CHECK(ccgp_multi_grid_marginal(m, X, N, D, y, 2.0, hyper, G, Nn, 50.0, 1, -1.0, o1, NULL, NULL) == 0 && same(o0, o1, G),
            "grid without argmax / logs");
      free(hyper); free(o0); free(o1); free(l0); free(l1);
    }
    /* prediction tables S x m */
    {
      const int S = 7, mt = 5;
      double* params = filled((size_t)S * P, 0.25);
      params[5] = -2.0;
      double* Xt = filled((size_t)mt * D, 0.7);
      double *m0 = filled((size_t)S * mt, 0), *v0 = filled((size_t)S * mt, 0), *m1 = filled((size_t)S * mt, 1),
             *v1 = filled((size_t)S * mt, 1), *be0 = filled(S, 0), *be1 = filled(S, 1);
      int st0[7], st1[7];
      const int r0 = ccgp_predict_batch(one, X, N, D, y, K, params, S, Xt, mt, 3.0, m0, v0, be0, st0);
      const int r1 = ccgp_multi_predict_batch(m, X, N, D, y, K, params, S, Xt, mt, 3.0, m1, v1, be1, st1);
      CHECK(r0 == 1 && r1 == 1, "predict return %d %d", r0, r1);
      CHECK(same(m0, m1, (size_t)S * mt) && same(v0, v1, (size_t)S * mt) && same(be0, be1, S) &&
                memcmp(st0, st1, sizeof st0) == 0, "predict w=%d", w);
      free(params); free(Xt); free(m0); free(v0); free(m1); free(v1); free(be0); free(be1);
    }
    CHECK(ccgp_multi_set_kernel(m, 1, 5.0) == CCGP_OK && ccgp_multi_set_kernel(m, 0, 0.0) == CCGP_OK, "set_kernel");
    CHECK(ccgp_multi_set_kernel(m, 1, 50.0) == CCGP_EINVAL, "set_kernel bad nu");
    ccgp_multi_destroy(m);
  }
  /* a failing shard (stub device 13): the negative code, the shard named, nothing written out of bounds */
  {
    const int devs[3] = {0, 13, 2};
    ccgp_multi* m = NULL;
    CHECK(ccgp_multi_create(3, devs, &m) == CCGP_OK, "multi_create with device 13");
    const int B = 10;
    double* params = filled((size_t)B * P, 0.5);
    double* ll = filled(B, 0);
    CHECK(ccgp_multi_loglik_batch(m, X, N, D, y, K, params, B, 1.0, 0, 0.0, ll, NULL, NULL) == CCGP_EHIP, "failing shard code");
    CHECK(strstr(ccgp_multi_last_error(m), "shard 1") != NULL, "failing shard message: %s", ccgp_multi_last_error(m));
    double o[4], hy[16];
    for (int i = 0; i < 16; ++i) hy[i] = 1.0 + i;
    CHECK(ccgp_multi_grid_marginal(m, X, N, D, y, 2.0, hy, 4, 3, 50.0, 1, -1.0, o, NULL, NULL) == CCGP_EHIP, "failing shard grid");
    free(params); free(ll);
    ccgp_multi_destroy(m);
    /* both shards failing at once */
    const int devs2[2] = {13, 13};
    CHECK(ccgp_multi_create(2, devs2, &m) == CCGP_OK, "multi_create 13,13");
    params = filled((size_t)B * P, 0.5);
    ll = filled(B, 0);
    CHECK(ccgp_multi_loglik_batch(m, X, N, D, y, K, params, B, 1.0, 0, 0.0, ll, NULL, NULL) == CCGP_EHIP, "two failing shards");
    free(params); free(ll);
    ccgp_multi_destroy(m);
    /* a device that cannot be opened: create fails, nothing leaks */
    const int devs3[3] = {0, 1, 99};
    CHECK(ccgp_multi_create(3, devs3, &m) == CCGP_EHIP && m == NULL, "multi_create with a missing device");
    CHECK(ccgp_multi_create(0, NULL, &m) == CCGP_EINVAL, "multi_create(0)");
  }
  ccgp_destroy(one);
  free(X); free(y);
}

static SEXP call(const char* name, int nargs, SEXP* a) {
  SEXP r = rmock_dot_call(name, nargs, a);
  CHECK(r != NULL, "%s: %s", name, rmock_last_error());
  return r;
}
static double* reals(SEXP x) { return (double*)rmock_data(x); }

static void shim_scenarios(void) {
  double* X = filled((size_t)N * D, 0.1);
  double* y = filled(N, 2.0);
  double* th = filled(D, 0.3);
  const int Kk = K, zero = 0, one_i = 1;
  CHECK(rmock_load() >= 20, "registration");
  rmock_reset();
  SEXP a[8];
  /* single device: every routine once */
  a[0] = rmock_real(X, N * D, N, D); a[1] = rmock_real(th, D, -1, 0);
  SEXP r = call("ccgp_R_corr_matrix", 2, a);
  CHECK(r && rmock_length(r) == N * N, "corr_matrix length");
  double* Xn = filled(2 * D, 0.9);
  a[0] = rmock_real(Xn, 2 * D, 2, D); a[1] = rmock_real(X, N * D, N, D); a[2] = rmock_real(th, D, -1, 0);
  r = call("ccgp_R_corr_cross", 3, a);
  CHECK(r && rmock_length(r) == 2 * N, "corr_cross length");
  double* row = filled(P, 0.4);
  a[0] = rmock_real(X, N * D, N, D); a[1] = rmock_int(&Kk, 1); a[2] = rmock_real(row, P, -1, 0);
  r = call("ccgp_R_mixed_corr_matrix", 3, a);
  const int K9 = 9;
  a[1] = rmock_int(&K9, 1);
  r = call("ccgp_R_mixed_corr_matrix", 3, a);                       /* negative code: warning + NA */
  CHECK(r && rmock_is_na_real(reals(r)[0]) && rmock_is_na_real(reals(r)[N * N - 1]) && rmock_n_warnings() == 1, "K = 9 -> NA");
  a[0] = rmock_real(Xn, 2 * D, 2, D); a[1] = rmock_real(X, N * D, N, D); a[2] = rmock_int(&Kk, 1); a[3] = rmock_real(row, P, -1, 0);
  call("ccgp_R_mixed_corr_cross", 4, a);
  double tt[3] = {0.1, 0.2, 0.3}, s2 = 2.5, pp[4] = {7, 3, 3, 28};
  a[0] = rmock_real(X, N * D, N, D); a[1] = rmock_real(tt, 3, -1, 0); a[2] = rmock_real(y, N, -1, 0); a[3] = rmock_real(&s2, 1, -1, 0);
  a[4] = rmock_int(&zero, 1); a[5] = rmock_real(pp, 4, -1, 0); a[6] = rmock_int(&one_i, 1);
  r = call("ccgp_R_logpost", 7, a);
  CHECK(r && rmock_length(r) == 4 && rmock_length(rmock_elt(r, 2)) == N * N, "logpost list");
  a[6] = rmock_int(&zero, 1);                                        /* slim frame: R.Inv is the 1 x 1 placeholder */
  r = call("ccgp_R_logpost", 7, a);
  CHECK(r && rmock_length(rmock_elt(r, 2)) == 1 && reals(rmock_elt(r, 2))[0] == 0.0, "logpost without R.Inv");
  tt[0] = -800.0;
  a[1] = rmock_real(tt, 3, -1, 0); a[4] = rmock_int(&one_i, 1); a[5] = rmock_nil(); a[6] = rmock_int(&one_i, 1);
  r = call("ccgp_R_logpost", 7, a);
  CHECK(r && rmock_is_na_real(reals(rmock_elt(r, 0))[0]) && rmock_length(rmock_elt(r, 2)) == 1, "failed logpost -> NA, R.Inv <- NA");
  {   /* speculative block of Metro: 2^5 - 1 candidates, R_alloc scratch, one failing candidate */
    double th0[3] = {0.1, -0.2, 0.3}, st0[2] = {-0.5, 1.0}, uu[5] = {0.9, 0.2, 0.6, 0.05, 0.5}, pr[5] = {0, 7, 3, 3, 28};
    double* EE = filled(5 * 3, 0.07);
    EE[2] = -900.0;                                                  /* proposal 3 of every history: factorisation "fails" */
    a[0] = rmock_real(X, N * D, N, D); a[1] = rmock_real(y, N, -1, 0); a[2] = rmock_real(&s2, 1, -1, 0); a[3] = rmock_real(pr, 5, -1, 0);
    a[4] = rmock_real(th0, 3, -1, 0); a[5] = rmock_real(st0, 2, -1, 0); a[6] = rmock_real(uu, 5, -1, 0); a[7] = rmock_real(EE, 15, 5, 3);
    r = call("ccgp_R_metro_steps", 8, a);
    CHECK(r && rmock_length(r) == 6 && rmock_length(rmock_elt(r, 1)) == 15 && rmock_is_na_real(reals(rmock_elt(r, 4))[2]), "metro steps");
    a[6] = rmock_real(uu, 0, -1, 0);
    r = rmock_dot_call("ccgp_R_metro_steps", 8, a);                   /* m = 0: Rf_error, not a crash */
    CHECK(!r, "metro steps refuses an empty block");
    free(EE);
  }
  const int B = 11;
  double* params = filled((size_t)B * P, 0.5);
  params[4] = -1.0;
  double tau2 = 4.0;
  a[0] = rmock_real(X, N * D, N, D); a[1] = rmock_real(y, N, -1, 0); a[2] = rmock_int(&Kk, 1); a[3] = rmock_real(params, B * P, B, P);
  a[4] = rmock_real(&s2, 1, -1, 0); a[5] = rmock_int(&one_i, 1); a[6] = rmock_real(&tau2, 1, -1, 0);
  SEXP single_ll = call("ccgp_R_loglik_batch", 7, a);
  CHECK(single_ll && rmock_is_na_real(reals(rmock_elt(single_ll, 0))[4]) && !rmock_is_na_real(reals(rmock_elt(single_ll, 0))[3]),
        "NaN -> NA_real_ for the failed evaluation only");
  double* ll_keep = (double*)malloc(sizeof(double) * B);
  memcpy(ll_keep, reals(rmock_elt(single_ll, 0)), sizeof(double) * B);
  double hy[20];
  for (int i = 0; i < 20; ++i) hy[i] = 1.0 + 0.5 * i;
  const int Nn = 6;
  double tau = 50.0, lam = -1.0;
  a[0] = rmock_real(X, N * D, N, D); a[1] = rmock_real(y, N, -1, 0); a[2] = rmock_real(&s2, 1, -1, 0); a[3] = rmock_real(hy, 20, 5, 4);
  a[4] = rmock_int(&Nn, 1); a[5] = rmock_real(&tau, 1, -1, 0); a[6] = rmock_int(&one_i, 1); a[7] = rmock_real(&lam, 1, -1, 0);
  SEXP single_grid = call("ccgp_R_grid_marginal", 8, a);
  double grid_keep[5];
  memcpy(grid_keep, reals(rmock_elt(single_grid, 0)), sizeof grid_keep);
  const int arg_keep = ((int*)rmock_data(rmock_elt(single_grid, 1)))[0];
  double* Xt = filled(4 * D, 0.7);
  a[0] = rmock_real(X, N * D, N, D); a[1] = rmock_real(y, N, -1, 0); a[2] = rmock_int(&Kk, 1); a[3] = rmock_real(params, B * P, B, P);
  a[4] = rmock_real(Xt, 4 * D, 4, D); a[5] = rmock_real(&s2, 1, -1, 0);
  SEXP single_pred = call("ccgp_R_predict_batch", 6, a);
  double* mean_keep = (double*)malloc(sizeof(double) * B * 4);
  memcpy(mean_keep, reals(rmock_elt(single_pred, 0)), sizeof(double) * B * 4);
  /* the prediction phase: a frame (data.frame = list of columns, or a matrix) -> one table; cache / lookup / clear */
  const int S = 5, mt = 4;
  double* fr = filled((size_t)S * 4, 0.2);                           /* p, theta1, theta2, beta columns */
  for (int i = 0; i < S; ++i) fr[i] = 0.1 + 0.15 * i;
  SEXP frame = rmock_list(4);
  for (int j = 0; j < 4; ++j) rmock_list_set(frame, j, rmock_real(fr + (size_t)j * S, S, -1, 0));
  const char* fn[4] = {"p", "theta1", "theta2", "beta"};
  rmock_list_names(frame, fn, 4);
  double zero_d = 0.0;
  a[0] = frame; a[1] = rmock_real(X, N * D, N, D); a[2] = rmock_real(Xt, mt * D, mt, D); a[3] = rmock_real(&s2, 1, -1, 0);
  a[4] = rmock_real(y, N, -1, 0); a[5] = rmock_int(&zero, 1); a[6] = rmock_real(&zero_d, 1, -1, 0);
  SEXP tab = call("ccgp_R_prediction_table", 7, a);
  CHECK(tab && rmock_length(tab) == 3 && rmock_length(rmock_elt(tab, 0)) == S * mt && rmock_length(rmock_elt(tab, 2)) == S, "table shapes");
  a[0] = rmock_real(fr, S * 4, S, 4);                               /* the same draws as a numeric matrix */
  SEXP tab2 = call("ccgp_R_prediction_table", 7, a);
  CHECK(tab2 && same(reals(rmock_elt(tab, 0)), reals(rmock_elt(tab2, 0)), (size_t)S * mt) &&
            same(reals(rmock_elt(tab, 1)), reals(rmock_elt(tab2, 1)), (size_t)S * mt), "data frame == matrix");
  a[0] = frame;
  r = call("ccgp_R_table_cache", 7, a);
  CHECK(r && ((int*)rmock_data(r))[0] == S, "table_cache returns the number of draws");
  for (int pass = 0; pass < 2; ++pass)                               /* in order, then again (wrap-around) */
    for (int t = 0; t < mt; ++t) {
      double xr[D];
      for (int k = 0; k < D; ++k) xr[k] = Xt[t + (size_t)k * mt];
      SEXP b[2] = {rmock_real(xr, D, -1, 0), rmock_int(&S, 1)};
      r = call("ccgp_R_table_lookup", 2, b);
      int ok = r && rmock_length(r) == 2 * S;
      for (int q = 0; ok && q < S; ++q)
        ok = reals(r)[2 * q] == reals(rmock_elt(tab, 0))[q + (size_t)t * S] && reals(r)[2 * q + 1] == reals(rmock_elt(tab, 1))[q + (size_t)t * S];
      CHECK(ok, "lookup of test site %d (pass %d)", t, pass);
    }
  {
    double far[D] = {9.0, 9.0, 9.0};
    const int S1 = S + 1;
    SEXP b[2] = {rmock_real(far, D, -1, 0), rmock_int(&S, 1)};
    r = call("ccgp_R_table_lookup", 2, b);
    CHECK(r && rmock_typeof(r) == 0, "unknown site -> NULL");
    double xr[D];
    for (int k = 0; k < D; ++k) xr[k] = Xt[(size_t)k * mt];
    b[0] = rmock_real(xr, D, -1, 0); b[1] = rmock_int(&S1, 1);
    r = call("ccgp_R_table_lookup", 2, b);
    CHECK(r && rmock_typeof(r) == 0, "another frame size -> NULL");
    call("ccgp_R_table_clear", 0, b);
    b[1] = rmock_int(&S, 1);
    r = call("ccgp_R_table_lookup", 2, b);
    CHECK(r && rmock_typeof(r) == 0, "cleared -> NULL");
  }
  /* a frame too narrow for the layout (ANI needs 4 leading columns): warning + NA, never an error */
  {
    const int two = 2, w0 = rmock_n_warnings();
    SEXP narrow = rmock_list(3);
    for (int j = 0; j < 3; ++j) rmock_list_set(narrow, j, rmock_real(fr + (size_t)j * S, S, -1, 0));
    double* X2 = filled((size_t)N * 2, 0.1);
    double* Xt2 = filled((size_t)mt * 2, 0.6);
    a[0] = narrow; a[1] = rmock_real(X2, N * 2, N, 2); a[2] = rmock_real(Xt2, mt * 2, mt, 2); a[5] = rmock_int(&two, 1);
    r = call("ccgp_R_prediction_table", 7, a);
    CHECK(r && rmock_is_na_real(reals(rmock_elt(r, 0))[0]) && rmock_n_warnings() == w0 + 1, "narrow frame -> warning + NA");
    free(X2); free(Xt2);
  }
  /* literal helpers */
  double* Rinv = filled((size_t)N * N, 0.01);
  double beta = 0.3;
  a[0] = rmock_real(Rinv, N * N, N, N); a[1] = rmock_real(&beta, 1, -1, 0); a[2] = rmock_real(y, N, -1, 0);
  r = call("ccgp_R_factors", 3, a);
  CHECK(r && rmock_length(r) == 2 * N + 1, "factors length");
  double* rr = filled(2 * N, 0.2);
  double vf2 = 1.7;
  a[0] = rmock_real(rr, 2 * N, 2, N); a[1] = rmock_real(&beta, 1, -1, 0); a[2] = rmock_real(y, N, -1, 0); a[3] = rmock_real(y, N, -1, 0);
  a[4] = rmock_real(&vf2, 1, -1, 0); a[5] = rmock_real(Rinv, N * N, N, N); a[6] = rmock_real(&s2, 1, -1, 0);
  r = call("ccgp_R_predict_from_factors", 7, a);
  CHECK(r && rmock_length(r) == 4, "predict_from_factors is m x 2");
  {
    /* predict.post on one full frame row; on a slim row: warning + NA */
    const int len = 3 + 2 + 2 * N + N * N, w0 = rmock_n_warnings();
    double* prow = filled(len, 0.3);
    double xr[D] = {0.2, 0.4, 0.6};
    a[0] = rmock_real(xr, D, -1, 0); a[1] = rmock_real(X, N * D, N, D); a[2] = rmock_real(prow, len, -1, 0);
    a[3] = rmock_real(&s2, 1, -1, 0); a[4] = rmock_int(&zero, 1); a[5] = rmock_real(&zero_d, 1, -1, 0);
    r = call("ccgp_R_predict_post", 6, a);
    CHECK(r && rmock_length(r) == 2 && !rmock_is_na_real(reals(r)[0]) && rmock_n_warnings() == w0, "predict.post -> cbind(mean, var)");
    a[2] = rmock_real(prow, 4, -1, 0);
    r = call("ccgp_R_predict_post", 6, a);
    CHECK(r && rmock_is_na_real(reals(r)[0]) && rmock_is_na_real(reals(r)[1]) && rmock_n_warnings() == w0 + 1, "slim row -> warning + NA");
    free(prow);
  }
  a[0] = rmock_real(Rinv, N * N, N, N); a[1] = rmock_real(y, N, -1, 0);
  call("ccgp_R_beta_mle", 2, a);
  a[2] = rmock_real(&beta, 1, -1, 0);
  call("ccgp_R_sigma2_mle", 3, a);
  const int nd = 4, dd = 2, Bd = 3;
  double* Xs = filled((size_t)nd * dd * Bd, 0.3);
  double* row2 = filled(2 + 2 * dd, 0.4);
  a[0] = rmock_real(Xs, nd * dd * Bd, nd * dd, Bd); a[1] = rmock_int(&nd, 1); a[2] = rmock_int(&dd, 1); a[3] = rmock_int(&Kk, 1);
  a[4] = rmock_real(row2, 2 + 2 * dd, -1, 0);
  r = call("ccgp_R_mixed_logdet_designs", 5, a);
  CHECK(r && rmock_length(r) == Bd, "logdet designs length");
  double nu = 5.0;
  a[0] = rmock_int(&one_i, 1); a[1] = rmock_real(&nu, 1, -1, 0);
  call("ccgp_R_set_kernel", 2, a);
  a[0] = rmock_int(&zero, 1);
  call("ccgp_R_set_kernel", 2, a);
  r = call("ccgp_R_devices", 0, a);
  CHECK(r && ((int*)rmock_data(r))[0] == 1, "one device");
  CHECK(rmock_counter(0) == 0 && rmock_counter(1) == 0 && rmock_counter(2) == 0 && rmock_counter(3) == 0 && rmock_counter(4) == 0,
        "mock bookkeeping: unbalanced %d hazards %d type %d underflow %d depth %d", rmock_counter(0), rmock_counter(1),
        rmock_counter(2), rmock_counter(3), rmock_counter(4));

  /* CCGP_DEVICES = "3": the batched routines go through ccgp_multi (devices 0, 1, 2), same bits */
  rmock_unload();
  setenv("CCGP_DEVICES", "3", 1);
  rmock_load();
  rmock_reset();
  r = call("ccgp_R_devices", 0, a);
  CHECK(r && ((int*)rmock_data(r))[0] == 3, "CCGP_DEVICES=3");
  a[0] = rmock_real(X, N * D, N, D); a[1] = rmock_real(y, N, -1, 0); a[2] = rmock_int(&Kk, 1); a[3] = rmock_real(params, B * P, B, P);
  a[4] = rmock_real(&s2, 1, -1, 0); a[5] = rmock_int(&one_i, 1); a[6] = rmock_real(&tau2, 1, -1, 0);
  r = call("ccgp_R_loglik_batch", 7, a);
  CHECK(r && same(reals(rmock_elt(r, 0)), ll_keep, B), "sharded loglik through the shim");
  a[0] = rmock_real(X, N * D, N, D); a[1] = rmock_real(y, N, -1, 0); a[2] = rmock_real(&s2, 1, -1, 0); a[3] = rmock_real(hy, 20, 5, 4);
  a[4] = rmock_int(&Nn, 1); a[5] = rmock_real(&tau, 1, -1, 0); a[6] = rmock_int(&one_i, 1); a[7] = rmock_real(&lam, 1, -1, 0);
  r = call("ccgp_R_grid_marginal", 8, a);
  CHECK(r && same(reals(rmock_elt(r, 0)), grid_keep, 5) && ((int*)rmock_data(rmock_elt(r, 1)))[0] == arg_keep, "sharded grid through the shim");
  a[0] = rmock_real(X, N * D, N, D); a[1] = rmock_real(y, N, -1, 0); a[2] = rmock_int(&Kk, 1); a[3] = rmock_real(params, B * P, B, P);
  a[4] = rmock_real(Xt, 4 * D, 4, D); a[5] = rmock_real(&s2, 1, -1, 0);
  r = call("ccgp_R_predict_batch", 6, a);
  CHECK(r && same(reals(rmock_elt(r, 0)), mean_keep, (size_t)B * 4), "sharded predict through the shim");
  a[0] = rmock_int(&one_i, 1); a[1] = rmock_real(&nu, 1, -1, 0);
  call("ccgp_R_set_kernel", 2, a);
  a[0] = rmock_int(&zero, 1);
  call("ccgp_R_set_kernel", 2, a);
  CHECK(rmock_n_warnings() == 0, "no warning on the sharded path: %s", rmock_warning(0));

  /* a list with a failing device: warning + NA, never an error */
  rmock_unload();
  setenv("CCGP_DEVICES", "0,13", 1);
  rmock_load();
  rmock_reset();
  a[0] = rmock_real(X, N * D, N, D); a[1] = rmock_real(y, N, -1, 0); a[2] = rmock_int(&Kk, 1); a[3] = rmock_real(params, B * P, B, P);
  a[4] = rmock_real(&s2, 1, -1, 0); a[5] = rmock_int(&one_i, 1); a[6] = rmock_real(&tau2, 1, -1, 0);
  r = call("ccgp_R_loglik_batch", 7, a);
  CHECK(r && rmock_is_na_real(reals(rmock_elt(r, 0))[0]) && rmock_is_na_real(reals(rmock_elt(r, 0))[B - 1]) &&
            rmock_n_warnings() == 1 && strstr(rmock_warning(0), "shard 1"), "failing shard -> warning + NA: %s", rmock_warning(0));
  /* a count beyond what exists: ccgp_multi_create fails, the shim warns and falls back to one device */
  rmock_unload();
  setenv("CCGP_DEVICES", "99", 1);
  rmock_load();
  rmock_reset();
  r = call("ccgp_R_devices", 0, a);
  CHECK(r && ((int*)rmock_data(r))[0] == 1 && rmock_n_warnings() == 1, "CCGP_DEVICES=99 falls back");
  /* malformed lists */
  const char* odd[] = {"", ",", "1,", "abc", "2,,3"};
  for (int i = 0; i < 5; ++i) {
    rmock_unload();
    setenv("CCGP_DEVICES", odd[i], 1);
    rmock_load();
    rmock_reset();
    r = call("ccgp_R_devices", 0, a);
    CHECK(r != NULL, "CCGP_DEVICES=\"%s\"", odd[i]);
  }
  rmock_unload();
  unsetenv("CCGP_DEVICES");
  rmock_reset();
  free(X); free(y); free(th); free(Xn); free(row); free(params); free(ll_keep); free(mean_keep); free(Xt); free(Rinv); free(rr);
  free(Xs); free(row2); free(fr);
}

int main(void) {
  multi_scenarios();
  shim_scenarios();
  if (failures) {
    fprintf(stderr, "host-sanitizer driver: %d check(s) failed\n", failures);
    return 1;
  }
  printf("host-sanitizer driver: OK\n");
  return 0;
}
