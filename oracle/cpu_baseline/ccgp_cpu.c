/*
 * CPU baseline evaluator -- TEST / BENCH INFRASTRUCTURE, not product code.
 *
 * "time the build's own CPU backend: OpenMP over evaluations, one evaluation per
 * core -- mirrors how an R user would mclapply the grid -- on the GPU box host, all cores and one core".
 * This file is that evaluator.  It is compiled into oracle/cpu_baseline/libccgp_cpu.so, loaded only by
 * bench.py's cpu_baseline leg and by tests/test_cpu_baseline.py (which checks it against the oracle);
 * it is never linked into libccgp.so and never imported by the package.
 *
 * Per evaluation (one hyperparameter draw), the algorithmic minimum the GPU path also performs:
 *   covariance build   Mixed.corr.matrix HX:408-415 / ANI:399-406 in the reference's expanded-distance
 *                      form (HX:352-355), lower triangle only
 *   Cholesky           LAPACK dpotrf (the factorisation inside mnormt::dmnorm, HX:460 / HX:570)
 *   two forward solves dtrsv on [y 1]  ->  beta.MLE (HX:384-388) and the quadratic form
 *   log-likelihood     dmnorm(..., log = TRUE)
 * and for prediction (predict.post HX:655-673) one more dtrsv per test site.
 * dpotrf / dtrsv come from the OpenBLAS that scipy bundles (single-threaded inside each evaluation), bound
 * at run time with dlopen so that nothing has to be installed; without it a plain blocked Cholesky in C
 * is used and the caller reports that.  The reference's own operation count (LU inverse + Cholesky +
 * chol2inv on materialised temporaries) is the separate "port" figure (oracle/ccgp_oracle.py).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

typedef void (*potrf_fn)(const char*, const int*, double*, const int*, int*);
typedef void (*trsv_fn)(const char*, const char*, const char*, const int*, const double*, const int*, double*,
                        const int*);
typedef void (*setthr_fn)(int);
typedef int (*getthr_fn)(void);

static potrf_fn g_potrf = 0;
static trsv_fn g_trsv = 0;
static setthr_fn g_setthr = 0;
static getthr_fn g_getthr = 0;

/* returns 1 when dpotrf / dtrsv were bound from `path`, 0 when the built-in factorisation will be used */
int ccgp_cpu_init(const char* path) {
  if (g_potrf) return 1;
  if (!path || !*path) return 0;
  void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!h) return 0;
  const char* pn[] = {"scipy_dpotrf_", "dpotrf_"};
  const char* tn[] = {"scipy_dtrsv_", "dtrsv_"};
  const char* sn[] = {"scipy_openblas_set_num_threads", "openblas_set_num_threads"};
  const char* gn[] = {"scipy_openblas_get_num_threads", "openblas_get_num_threads"};
  potrf_fn p = 0;
  trsv_fn t = 0;
  setthr_fn s = 0;
  getthr_fn g = 0;
  for (int i = 0; i < 2 && !p; ++i) p = (potrf_fn)dlsym(h, pn[i]);
  for (int i = 0; i < 2 && !t; ++i) t = (trsv_fn)dlsym(h, tn[i]);
  for (int i = 0; i < 2 && !s; ++i) s = (setthr_fn)dlsym(h, sn[i]);
  for (int i = 0; i < 2 && !g; ++i) g = (getthr_fn)dlsym(h, gn[i]);
  /* One evaluation per core means no threading INSIDE an evaluation.  The library instance is shared with
   * scipy in this process, so its thread count is set to 1 by the CALLING thread around each batch and put
   * back afterwards (blas_single / blas_restore); without both hooks the built-in factorisation is used */
  if (!p || !t || !s || !g) return 0;
  g_potrf = p;
  g_trsv = t;
  g_setthr = s;
  g_getthr = g;
  return 1;
}

static int blas_single(void) {
  if (!g_setthr) return 0;
  const int prev = g_getthr();
  g_setthr(1);
  return prev;
}
static void blas_restore(int prev) {
  if (g_setthr && prev > 0) g_setthr(prev);
}

/* plain blocked Cholesky (lower, column-major), used only when no LAPACK could be bound.  ptol: a pivot <= ptol fails
 * (0 for the grid likelihood, n * DBL_EPSILON for the profile-beta likelihood whose reference path calls solve(R), which
 * refuses rcond < eps, HX:454 -- the same rule as the device, csrc/ccgp_internal.h pivot_tolerance) */
static int chol_builtin(double* A, int n, double ptol) {
  const int NB = 64;
  for (int j0 = 0; j0 < n; j0 += NB) {
    const int jb = n - j0 < NB ? n - j0 : NB;
    for (int j = j0; j < j0 + jb; ++j) {
      for (int k = j0; k < j; ++k) {
        const double l = A[j + (size_t)k * n];
        for (int i = j; i < n; ++i) A[i + (size_t)j * n] -= A[i + (size_t)k * n] * l;
      }
      const double piv = A[j + (size_t)j * n];
      if (!(piv > ptol)) return j + 1;
      const double s = sqrt(piv), rs = 1.0 / s;
      A[j + (size_t)j * n] = s;
      for (int i = j + 1; i < n; ++i) A[i + (size_t)j * n] *= rs;
    }
    /* trailing update with the finished block column */
    for (int c = j0 + jb; c < n; ++c)
      for (int k = j0; k < j0 + jb; ++k) {
        const double l = A[c + (size_t)k * n];
        for (int i = c; i < n; ++i) A[i + (size_t)c * n] -= A[i + (size_t)k * n] * l;
      }
  }
  return 0;
}

static void fwd_builtin(const double* L, int n, double* x) {
  for (int j = 0; j < n; ++j) {
    const double v = x[j] / L[j + (size_t)j * n];
    x[j] = v;
    for (int i = j + 1; i < n; ++i) x[i] -= L[i + (size_t)j * n] * v;
  }
}

/* n <= kSmallN: the built-in routines -- concurrent tiny LAPACK calls serialise on OpenBLAS's buffer lock
 * (measured: 8 threads slower than 1 at n = 64), and a 64 x 64 factorisation needs no blocking */
enum { kSmallN = 128 };

static int factor(double* A, int n, double ptol) {
  if (g_potrf && n > kSmallN) {
    int info = 0;
    g_potrf("L", &n, A, &n, &info);
    if (info == 0 && ptol > 0.0)
      for (int j = 0; j < n; ++j)
        if (!(A[j + (size_t)j * n] * A[j + (size_t)j * n] > ptol)) return j + 1;
    return info;
  }
  return chol_builtin(A, n, ptol);
}

static void fwd(const double* L, int n, double* x) {
  if (g_trsv && n > kSmallN) {
    const int one = 1;
    g_trsv("L", "N", "N", &n, L, &n, x, &one);
  } else {
    fwd_builtin(L, n, x);
  }
}

/* lower triangle of scale * R_mixed + shift; xt = X scaled per component: xt[c][k][i] = theta_ck x_ik */
static void build_cov_inner(const double* X, int n, int d, int K, const double* row, int ldp, double scale, double shift,
                            double* A, double* u, double* xt, int inner);
static void build_cov(const double* X, int n, int d, int K, const double* row, int ldp, double scale, double shift,
                      double* A, double* u, double* xt) {
  build_cov_inner(X, n, d, K, row, ldp, scale, shift, A, u, xt, 1);
}
/* inner > 1: the columns of ONE matrix are shared among `inner` OpenMP threads (an evaluation that has several cores to
 * itself: the concurrency x BLAS-threads sweep of bench.py); the scratch row then lives per thread */
static void build_cov_inner(const double* X, int n, int d, int K, const double* row, int ldp, double scale, double shift,
                            double* A, double* u, double* xt, int inner) {
  double sw = 0.0;
  for (int c = 0; c < K; ++c) sw += row[(size_t)c * ldp] * row[(size_t)c * ldp];
  for (int c = 0; c < K; ++c)
    for (int i = 0; i < n; ++i) {
      double s = 0.0;
      for (int k = 0; k < d; ++k) {
        const double th = row[(size_t)(K + c * d + k) * ldp], x = X[i + (size_t)k * n];
        s += x * x * th;
        xt[((size_t)c * d + k) * n + i] = th * x;
      }
      u[(size_t)c * n + i] = s;
    }
  const double f = scale / sw;
#pragma omp parallel num_threads(inner) if (inner > 1)
  {
  double* tmp = inner > 1 ? (double*)malloc(sizeof(double) * n) : xt + (size_t)K * d * n;   /* n doubles of scratch (behind the scaled coordinates when single-threaded) */
#pragma omp for schedule(dynamic, 8)
  for (int j = 0; j < n; ++j) {
    double* col = A + (size_t)j * n;
    for (int i = j; i < n; ++i) col[i] = shift;
    for (int c = 0; c < K; ++c) {
      const double w2 = row[(size_t)c * ldp] * row[(size_t)c * ldp] * f;
      const double uj = u[(size_t)c * n + j];
      const double* uc = u + (size_t)c * n;
      for (int i = j; i < n; ++i) tmp[i] = -(uc[i] + uj);
      for (int k = 0; k < d; ++k) {
        const double xj2 = 2.0 * X[j + (size_t)k * n];
        const double* xc = xt + ((size_t)c * d + k) * n;
        for (int i = j; i < n; ++i) tmp[i] += xc[i] * xj2;
      }
      for (int i = j; i < n; ++i) col[i] += w2 * exp(tmp[i]);
    }
  }
  if (inner > 1) free(tmp);
  }
}

static double sum_w2(const double* row, int K, int ldp) {
  double sw = 0.0;
  for (int c = 0; c < K; ++c) sw += row[(size_t)c * ldp] * row[(size_t)c * ldp];
  return sw;
}

static const double kLog2Pi = 1.8378770664093454835606594728112;

/* params: B x P column-major with leading dimension ldp (element (b, j) at params[b + j*ldp]).
 * mode 0: dmnorm(y, beta_hat, sigma2 sum(w^2) R);  mode 1: dmnorm(y, 0, sigma2 sum(w^2) R + tau2 11').
 * Returns the number of evaluations whose factorisation failed (status[b] != 0, loglik NaN). */
int ccgp_cpu_loglik_batch(const double* X, int n, int d, const double* y, int K, const double* params, int ldp, int B,
                          double sigma2, int mode, double tau2, double* out_ll, double* out_beta, int* status,
                          int threads) {
  int bad = 0;
  if (threads < 1) threads = omp_get_max_threads();
  const int prev_blas = blas_single();
#pragma omp parallel num_threads(threads) reduction(+ : bad)
  {
    double* A = (double*)malloc(sizeof(double) * (size_t)n * n);
    double* u = (double*)malloc(sizeof(double) * (size_t)K * n);
    double* xt = (double*)malloc(sizeof(double) * ((size_t)K * d + 1) * n);
    double* zy = (double*)malloc(sizeof(double) * n);
    double* z1 = (double*)malloc(sizeof(double) * n);
#pragma omp for schedule(dynamic)
    for (int b = 0; b < B; ++b) {
      const double* row = params + b;
      const double cs = sigma2 * sum_w2(row, K, ldp);
      build_cov(X, n, d, K, row, ldp, mode == 1 ? cs : 1.0, mode == 1 ? tau2 : 0.0, A, u, xt);
      const int info = factor(A, n, mode == 0 ? n * 2.220446049250313e-16 : 0.0);
      if (info != 0) {
        if (status) status[b] = info;
        out_ll[b] = NAN;
        if (out_beta) out_beta[b] = NAN;
        ++bad;
        continue;
      }
      if (status) status[b] = 0;
      double logdet = 0.0;
      for (int i = 0; i < n; ++i) logdet += log(A[i + (size_t)i * n]);
      logdet *= 2.0;
      memcpy(zy, y, sizeof(double) * n);
      fwd(A, n, zy);
      double ll, beta = 0.0;
      if (mode == 0) {
        for (int i = 0; i < n; ++i) z1[i] = 1.0;
        fwd(A, n, z1);
        double s11 = 0.0, s1y = 0.0;
        for (int i = 0; i < n; ++i) { s11 += z1[i] * z1[i]; s1y += z1[i] * zy[i]; }
        beta = s1y / s11;
        double q = 0.0;
        for (int i = 0; i < n; ++i) { const double v = zy[i] - beta * z1[i]; q += v * v; }
        ll = -0.5 * (n * kLog2Pi + n * log(cs) + logdet + q / cs);
      } else {
        double q = 0.0;
        for (int i = 0; i < n; ++i) q += zy[i] * zy[i];
        ll = -0.5 * (n * kLog2Pi + logdet + q);
      }
      out_ll[b] = ll;
      if (out_beta) out_beta[b] = beta;
    }
    free(A); free(u); free(xt); free(zy); free(z1);
  }
  blas_restore(prev_blas);
  return bad;
}

/* predict.post for S draws x m test sites: mean / var are S x m column-major */
int ccgp_cpu_predict_batch(const double* X, int n, int d, const double* y, int K, const double* params, int ldp, int S,
                           const double* Xt, int m, double sigma2, double* mean, double* var, int threads) {
  int bad = 0;
  if (threads < 1) threads = omp_get_max_threads();
  const int prev_blas = blas_single();
#pragma omp parallel num_threads(threads) reduction(+ : bad)
  {
    double* A = (double*)malloc(sizeof(double) * (size_t)n * n);
    double* u = (double*)malloc(sizeof(double) * (size_t)K * n);
    double* xt = (double*)malloc(sizeof(double) * ((size_t)K * d + 1) * n);
    double* zy = (double*)malloc(sizeof(double) * n);
    double* z1 = (double*)malloc(sizeof(double) * n);
    double* r = (double*)malloc(sizeof(double) * n);
#pragma omp for schedule(dynamic, 1)
    for (int s = 0; s < S; ++s) {
      const double* row = params + s;
      const double sw = sum_w2(row, K, ldp);
      build_cov(X, n, d, K, row, ldp, 1.0, 0.0, A, u, xt);
      if (factor(A, n, n * 2.220446049250313e-16) != 0) {
        for (int t = 0; t < m; ++t) mean[s + (size_t)t * S] = var[s + (size_t)t * S] = NAN;
        ++bad;
        continue;
      }
      memcpy(zy, y, sizeof(double) * n);
      fwd(A, n, zy);
      for (int i = 0; i < n; ++i) z1[i] = 1.0;
      fwd(A, n, z1);
      double s11 = 0.0, s1y = 0.0;
      for (int i = 0; i < n; ++i) { s11 += z1[i] * z1[i]; s1y += z1[i] * zy[i]; }
      const double beta = s1y / s11;
      for (int t = 0; t < m; ++t) {
        for (int i = 0; i < n; ++i) r[i] = 0.0;
        for (int c = 0; c < K; ++c) {
          double ut = 0.0;
          for (int k = 0; k < d; ++k) {
            const double x = Xt[t + (size_t)k * m];
            ut += x * x * row[(size_t)(K + c * d + k) * ldp];
          }
          const double w2 = row[(size_t)c * ldp] * row[(size_t)c * ldp] / sw;
          for (int i = 0; i < n; ++i) {
            double sdot = 0.0;
            for (int k = 0; k < d; ++k) sdot += xt[((size_t)c * d + k) * n + i] * Xt[t + (size_t)k * m];
            r[i] += w2 * exp(-((ut - 2.0 * sdot) + u[(size_t)c * n + i]));
          }
        }
        fwd(A, n, r);
        double ww = 0.0, z1w = 0.0, zyw = 0.0;
        for (int i = 0; i < n; ++i) { ww += r[i] * r[i]; z1w += z1[i] * r[i]; zyw += zy[i] * r[i]; }
        const double uu = 1.0 - z1w;
        mean[s + (size_t)t * S] = beta + (zyw - beta * z1w);
        var[s + (size_t)t * S] = sigma2 * (1.0 - ww + uu * uu / s11);
      }
    }
    free(A); free(u); free(xt); free(zy); free(z1); free(r);
  }
  blas_restore(prev_blas);
  return bad;
}

/* ONE evaluation at a time with `inner` threads to itself (covariance columns shared by OpenMP, dpotrf / dtrsv with `inner`
 * OpenBLAS threads): the building block of bench.py's (concurrent evaluations x threads per evaluation) sweep, where the
 * concurrency comes from separate worker PROCESSES -- scipy's pthread OpenBLAS serialises multi-threaded calls that arrive
 * from several threads of one process.  mode / outputs as ccgp_cpu_loglik_batch. */
int ccgp_cpu_loglik_seq(const double* X, int n, int d, const double* y, int K, const double* params, int ldp, int B,
                        double sigma2, int mode, double tau2, double* out_ll, double* out_beta, int* status, int inner) {
  int bad = 0;
  if (inner < 1) inner = 1;
  const int prev = g_setthr ? g_getthr() : 0;
  if (g_setthr) g_setthr(inner);
  double* A = (double*)malloc(sizeof(double) * (size_t)n * n);
  double* u = (double*)malloc(sizeof(double) * (size_t)K * n);
  double* xt = (double*)malloc(sizeof(double) * ((size_t)K * d + 1) * n);
  double* zy = (double*)malloc(sizeof(double) * n);
  double* z1 = (double*)malloc(sizeof(double) * n);
  for (int b = 0; b < B; ++b) {
    const double* row = params + b;
    const double cs = sigma2 * sum_w2(row, K, ldp);
    build_cov_inner(X, n, d, K, row, ldp, mode == 1 ? cs : 1.0, mode == 1 ? tau2 : 0.0, A, u, xt, inner);
    const int info = factor(A, n, mode == 0 ? n * 2.220446049250313e-16 : 0.0);
    if (status) status[b] = info;
    if (info != 0) {
      out_ll[b] = NAN;
      if (out_beta) out_beta[b] = NAN;
      ++bad;
      continue;
    }
    double logdet = 0.0;
    for (int i = 0; i < n; ++i) logdet += log(A[i + (size_t)i * n]);
    logdet *= 2.0;
    memcpy(zy, y, sizeof(double) * n);
    fwd(A, n, zy);
    double ll, beta = 0.0, q = 0.0;
    if (mode == 0) {
      for (int i = 0; i < n; ++i) z1[i] = 1.0;
      fwd(A, n, z1);
      double s11 = 0.0, s1y = 0.0;
      for (int i = 0; i < n; ++i) { s11 += z1[i] * z1[i]; s1y += z1[i] * zy[i]; }
      beta = s1y / s11;
      for (int i = 0; i < n; ++i) { const double v = zy[i] - beta * z1[i]; q += v * v; }
      ll = -0.5 * (n * kLog2Pi + n * log(cs) + logdet + q / cs);
    } else {
      for (int i = 0; i < n; ++i) q += zy[i] * zy[i];
      ll = -0.5 * (n * kLog2Pi + logdet + q);
    }
    out_ll[b] = ll;
    if (out_beta) out_beta[b] = beta;
  }
  free(A); free(u); free(xt); free(zy); free(z1);
  if (g_setthr && prev > 0) g_setthr(prev);
  return bad;
}

/* ONE literal predict.post call with the cached terms given (HX:655-673): r = Mixed.corr.vec(x.new, D.train, ...) in the
 * expanded form (HX:367-375, HX:425-431), mean = beta + mean.factor' r, var = sigma2 (1 - r' R.Inv r + (1 - v1' r)^2 / v2).
 * What one R-level call costs in compiled code on one core: the sequential caller's CPU comparator in bench.py. */
void ccgp_cpu_predict_post(const double* x, const double* X, int n, int d, int K, const double* row, double beta,
                           const double* mean_factor, const double* v1, double v2, const double* Rinv, double sigma2,
                           double* r, double* out) {
  const double sw = sum_w2(row, K, 1);
  for (int i = 0; i < n; ++i) r[i] = 0.0;
  for (int c = 0; c < K; ++c) {
    double ut = 0.0;
    for (int k = 0; k < d; ++k) ut += x[k] * x[k] * row[K + c * d + k];
    const double w2 = row[c] * row[c] / sw;
    for (int i = 0; i < n; ++i) {
      double sdot = 0.0, ui = 0.0;
      for (int k = 0; k < d; ++k) {
        const double th = row[K + c * d + k], xi = X[i + (size_t)k * n];
        sdot += th * xi * x[k];
        ui += th * xi * xi;
      }
      r[i] += w2 * exp(-((ut - 2.0 * sdot) + ui));
    }
  }
  double mr = 0.0, vr = 0.0, q = 0.0;
  for (int j = 0; j < n; ++j) {
    const double* col = Rinv + (size_t)j * n;
    double t = 0.0;
    for (int i = 0; i < n; ++i) t += col[i] * r[i];
    q += t * r[j];
    mr += mean_factor[j] * r[j];
    vr += v1[j] * r[j];
  }
  out[0] = beta + mr;
  out[1] = sigma2 * (1.0 - q + (1.0 - vr) * (1.0 - vr) / v2);
}

int ccgp_cpu_max_threads(void) { return omp_get_max_threads(); }
