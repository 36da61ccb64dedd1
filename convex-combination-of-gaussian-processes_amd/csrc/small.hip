// Fused in-LDS evaluator for n <= 128 (BASELINE configs 2, 3, 5).
//
// One workgroup per (draw [, chunk of right-hand sides]).  The mixed covariance is
// built straight into LDS from X (never touching HBM), then factorised in place as
// A = L' D L'^T with the right-hand sides appended as EXTRA ROWS of the lower
// triangle:
//
//      rows 0..n-1   : the covariance (lower triangle)
//      row  n        : y'          row n+1 : 1'
//      rows n+2..    : chunk-specific rows (cross-correlations r(x_t)' for prediction,
//                      unit vectors e_t' for the explicit inverse / gradient)
//
// Eliminating column k updates those rows exactly like matrix rows, so when the
// factorisation ends each extra row holds z' = L'^-1 b (forward substitution for free)
// and every quantity the reference needs is a D-weighted dot product of rows:
//   logdet = sum log d_k,  b' A^-1 c = sum_k z'_b[k] z'_c[k] / d_k.
// Replaces, per draw: Mixed.corr.matrix (HX:408-415), solve(R) (HX:454), beta.MLE
// (HX:458), dmnorm (HX:460 / HX:570), factors (HX:604-613), predict.post (HX:655-673).
//
// Bound: neither HBM nor MFMA -- an n-step dependent chain (one barrier per column) over
// LDS-resident data; algorithmic HBM traffic is the parameter row in and a few doubles out.
#include "ccgp_internal.h"

namespace ccgp {

namespace {

enum { kRowsLoglik = 0, kRowsPredict = 1, kRowsUnit = 2 };

struct SmallArgs {
  const double* X;
  const double* y;
  int n, d;
  const double* params;
  int ldp, K;
  int draw0;
  double sigma2;
  int mode;
  double tau2;
  // chunked extra rows
  int kind;          // kRows*
  const double* Xt;  // m x d test sites (kRowsPredict)
  int m;             // total extra rows wanted (test points, or n for unit vectors)
  int mtile;         // extra rows per workgroup
  int S;             // number of draws (leading dimension of mean/var)
  // outputs
  double* loglik;
  double* beta;
  int* status;
  double* mean;
  double* var;
  double* Rinv;      // n x n (kRowsUnit, inverse)
  double* gpart;     // gradient partials [draw][chunk][P]
  int want_grad;
};

__device__ inline double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return __shfl(v, 0, 64);
}

// LDS carve (doubles):  A[ld*n] | xs[d*n] | us[K*n] | xt[d*mtile] | ut[K*mtile] | th[K*d] | w2[K] | red[16] | etab[256]
__global__ __launch_bounds__(256) void small_kernel(SmallArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int n = a.n, d = a.d, K = a.K;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = a.draw0 + blockIdx.x;
  const int chunk = blockIdx.y;
  const int t0 = chunk * a.mtile;
  const int mt = a.kind == kRowsLoglik ? 0 : min(a.mtile, a.m - t0);
  const int Rt = n + 2 + mt;
  const int ld = n + 2 + a.mtile;

  double* A = smem;
  double* xs = A + (size_t)ld * n;
  double* us = xs + d * n;
  double* xt = us + K * n;
  double* ut = xt + d * a.mtile;
  double* th = ut + K * a.mtile;
  double* w2 = th + K * d;
  double* red = w2 + K;
  double* etab = red + 16;   // 2^(j/256) for exp_cov

  if (CCGP_SMALL_EXP_TABLE) exp_table_load(etab, tid, 256);
  for (int e = tid; e < K * d; e += 256) th[e] = a.params[b + (size_t)(K + e) * a.ldp];
  if (tid < K) {
    double w = a.params[b + (size_t)tid * a.ldp];
    w2[tid] = w * w;
  }
  for (int e = tid; e < n * d; e += 256) xs[e] = a.X[e];  // xs[k*n + i]
  if (a.kind == kRowsPredict)
    for (int e = tid; e < mt * d; e += 256) {
      int k = e / mt, t = e % mt;
      xt[k * a.mtile + t] = a.Xt[(t0 + t) + (size_t)k * a.m];
    }
  __syncthreads();
  for (int e = tid; e < K * n; e += 256) {
    int c = e / n, i = e % n;
    double s = 0.0;
    for (int k = 0; k < d; ++k) { double v = xs[k * n + i]; s += v * v * th[c * d + k]; }
    us[e] = s;
  }
  if (a.kind == kRowsPredict)
    for (int e = tid; e < K * mt; e += 256) {
      int c = e / mt, t = e % mt;
      double s = 0.0;
      for (int k = 0; k < d; ++k) { double v = xt[k * a.mtile + t]; s += v * v * th[c * d + k]; }
      ut[c * a.mtile + t] = s;
    }
  __syncthreads();

  double sw = 0.0;
  for (int c = 0; c < K; ++c) sw += w2[c];
  const double cs = a.sigma2 * sw;
  const double post_scale = a.mode == 1 ? cs : 1.0;
  const double post_shift = a.mode == 1 ? a.tau2 : 0.0;

  // ---- fill: column j, rows j..Rt-1 -------------------------------------------------
  for (int j = wave; j < n; j += 4) {
    for (int i = j + lane; i < Rt; i += 64) {
      double v;
      if (i < n) {
        double acc = 0.0;
        for (int c = 0; c < K; ++c) {
          double s = 0.0;
          for (int k = 0; k < d; ++k) s = fma(xs[k * n + i] * th[c * d + k], xs[k * n + j], s);
          double dist = (us[c * n + i] + us[c * n + j]) + (-2.0 * s);
          acc += w2[c] * exp_small(dist, etab);
        }
        v = post_scale * (acc / sw) + post_shift;
      } else if (i == n) {
        v = a.y[j];
      } else if (i == n + 1) {
        v = 1.0;
      } else if (a.kind == kRowsPredict) {
        const int t = i - n - 2;
        double acc = 0.0;
        for (int c = 0; c < K; ++c) {
          double s = 0.0;
          for (int k = 0; k < d; ++k) s = fma(xs[k * n + j] * th[c * d + k], xt[k * a.mtile + t], s);
          // corr.vec order: (theta'x^2 - 2 X Theta x) + u_i   (HX:373)
          double dist = (ut[c * a.mtile + t] - 2.0 * s) + us[c * n + j];
          acc += w2[c] * exp_small(dist, etab);
        }
        v = acc / sw;
      } else {
        v = (i - n - 2 + t0 == j) ? 1.0 : 0.0;
      }
      A[i + (size_t)j * ld] = v;
    }
  }

  // ---- A = L' D L'^T, one barrier per column ------------------------------------------
  int bad = 0;
  for (int k = 0; k < n; ++k) {
    __syncthreads();
    const double piv = A[k + (size_t)k * ld];
    if (!(piv > pivot_tolerance(a.mode, n))) { bad = k + 1; break; }   // uniform: every thread reads the same word
    const double rinv = 1.0 / piv;
    const double* colk = A + (size_t)k * ld;
    for (int j = k + 1 + wave; j < n; j += 4) {
      const double ljk = colk[j] * rinv;
      double* colj = A + (size_t)j * ld;
      for (int i = j + lane; i < Rt; i += 64) colj[i] = fma(-colk[i], ljk, colj[i]);
    }
  }
  __syncthreads();

  // ---- reductions over the pivots (wave 0) ----------------------------------------------
  const double kNaN = __longlong_as_double(0x7ff8000000000000LL);
  if (wave == 0) {
    double logdet = 0.0, s11 = 0.0, s1y = 0.0, syy = 0.0;
    if (!bad) {
      for (int k = lane; k < n; k += 64) {
        double dk = A[k + (size_t)k * ld];
        double zy = A[n + (size_t)k * ld], z1 = A[n + 1 + (size_t)k * ld];
        logdet += log(dk);
        s11 += z1 * z1 / dk;
        s1y += z1 * zy / dk;
        syy += zy * zy / dk;
      }
    }
    logdet = wave_sum(logdet); s11 = wave_sum(s11); s1y = wave_sum(s1y); syy = wave_sum(syy);
    double beta = 0.0, quad;
    const double kLog2Pi = 1.8378770664093454835606594728112;
    double ll;
    if (a.mode == 0) {
      beta = s1y / s11;
      double q = 0.0;
      if (!bad)
        for (int k = lane; k < n; k += 64) {
          double dk = A[k + (size_t)k * ld];
          double r = A[n + (size_t)k * ld] - beta * A[n + 1 + (size_t)k * ld];
          q += r * r / dk;
        }
      quad = wave_sum(q);
      ll = -0.5 * (n * kLog2Pi + n * log(cs) + logdet + quad / cs);
    } else {
      quad = syy;
      ll = -0.5 * (n * kLog2Pi + logdet + quad);
    }
    if (bad) { ll = kNaN; beta = kNaN; }
    if (lane == 0) {
      red[0] = beta; red[1] = s11; red[2] = ll;
      if (chunk == 0) {
        if (a.loglik) a.loglik[b] = ll;
        if (a.beta) a.beta[b] = beta;
        if (a.status) a.status[b] = bad;
      }
    }
  }
  __syncthreads();
  if (a.kind == kRowsLoglik) return;

  const double beta = red[0], s11 = red[1];
  if (a.kind == kRowsPredict) {
    // mean = beta + (z_y - beta z_1).w ; var = sigma2 (1 - w.w + (1 - z_1.w)^2 / (z_1.z_1))
    for (int t = tid; t < mt; t += 256) {
      double ww = 0.0, z1w = 0.0, zyw = 0.0;
      if (!bad)
        for (int k = 0; k < n; ++k) {
          const double* col = A + (size_t)k * ld;
          double rd = 1.0 / col[k];
          double w = col[n + 2 + t];
          ww = fma(w * rd, w, ww);
          z1w = fma(col[n + 1] * rd, w, z1w);
          zyw = fma(col[n] * rd, w, zyw);
        }
      double mean = beta + (zyw - beta * z1w);
      double u = 1.0 - z1w;
      double var = a.sigma2 * (1.0 - ww + u * u / s11);
      if (bad) { mean = kNaN; var = kNaN; }
      a.mean[b + (size_t)(t0 + t) * a.S] = mean;
      a.var[b + (size_t)(t0 + t) * a.S] = var;
    }
    return;
  }

  // ---- kRowsUnit: back-substitute rows y, 1 and the unit rows: x = L'^-T D^-1 z' ---------
  // thread r owns extra row n+r (r = 0: y, 1: ones, 2..: e_{t0+r-2}); in place, k downward.
  {
    const int nrows = 2 + mt;
    for (int r = tid; r < nrows; r += 256) {
      if (bad) break;
      const int row = n + r;
      for (int k = n - 1; k >= 0; --k) {
        const double* col = A + (size_t)k * ld;
        double rd = 1.0 / col[k];
        double acc = A[row + (size_t)k * ld];
        for (int i = k + 1; i < n; ++i) acc = fma(-col[i], A[row + (size_t)i * ld], acc);
        A[row + (size_t)k * ld] = acc * rd;
      }
    }
  }
  __syncthreads();
  if (a.Rinv) {
    for (int e = tid; e < mt * n; e += 256) {
      int t = e / n, i = e % n;
      a.Rinv[i + (size_t)(t0 + t) * n] = bad ? kNaN : A[n + 2 + t + (size_t)i * ld];
    }
  }
  if (a.want_grad) {
    // M[i][t] = 0.5 (alpha_i alpha_t - Sigma^-1[i][t]),  Sigma = cs * R,
    // alpha = R^-1 (y - beta 1) / cs.  Partial sums over this chunk's columns t:
    //   G_c  = sum M R_c ,  H_ck = sum M D_k R_c      (R_c recomputed, never stored)
    const int P = K + K * d;
    double* gp = a.gpart + ((size_t)blockIdx.x * gridDim.y + chunk) * P;
    for (int q = wave; q < P; q += 4) {
      const int c = q < K ? q : (q - K) / d;
      const int kk = q < K ? -1 : (q - K) % d;
      double acc = 0.0;
      if (!bad)
        for (int e = lane; e < mt * n; e += 64) {
          int t = e / n, i = e % n;
          int jt = t0 + t;
          double ai = (A[n + (size_t)i * ld] - beta * A[n + 1 + (size_t)i * ld]) / cs;
          double at = (A[n + (size_t)jt * ld] - beta * A[n + 1 + (size_t)jt * ld]) / cs;
          double Mit = 0.5 * (ai * at - A[n + 2 + t + (size_t)i * ld] / cs);
          double s = 0.0;
          for (int k = 0; k < d; ++k) s = fma(xs[k * n + i] * th[c * d + k], xs[k * n + jt], s);
          double dist = (us[c * n + i] + us[c * n + jt]) + (-2.0 * s);
          double rc = exp_small(dist, etab);
          if (kk >= 0) { double df = xs[kk * n + i] - xs[kk * n + jt]; rc *= df * df; }
          acc = fma(Mit, rc, acc);
        }
      acc = wave_sum(acc);
      if (lane == 0) {
        double wc = a.params[b + (size_t)c * a.ldp];
        double g = kk < 0 ? 2.0 * a.sigma2 * wc * acc : -a.sigma2 * wc * wc * acc;
        gp[q] = bad ? kNaN : g;
      }
    }
  }
}

__global__ void grad_reduce_kernel(const double* gpart, int nchunks, int P, int B, double* grad) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * P) return;
  int b = idx / P, q = idx % P;
  double s = 0.0;
  for (int c = 0; c < nchunks; ++c) s += gpart[((size_t)b * nchunks + c) * P + q];
  grad[b + (size_t)q * B] = s;
}

}  // namespace

size_t small_lds_bytes(int n, int d, int mtile) {
  size_t dbl = (size_t)(n + 2 + mtile) * n + (size_t)d * n + (size_t)kMaxK * n +
               (size_t)d * mtile + (size_t)kMaxK * mtile + (size_t)kMaxK * d + kMaxK + 16 + kExpTableDoubles;
  return dbl * sizeof(double);
}

int small_pick_mtile(int n, int d, int m) {
  const size_t budget = 150 * 1024;
  int mt = m < 1 ? 1 : m;
  if (mt > 256) mt = 256;
  while (mt > 1 && small_lds_bytes(n, d, mt) > budget) --mt;
  return mt;
}

static void small_launch(hipStream_t s, const SmallArgs& a, int ndraws, int nchunks) {
  size_t lds = small_lds_bytes(a.n, a.d, a.kind == kRowsLoglik ? 0 : a.mtile);
  static unsigned long long attr_mask = 0;
  once_per_device(attr_mask, [] {
    raise_lds_limit((const void*)small_kernel, "small_kernel");
  });
  hipLaunchKernelGGL(small_kernel, dim3(ndraws, nchunks), dim3(256), lds, s, a);
}

void launch_small_loglik(hipStream_t s, const double* X, int n, int d, const double* y, DrawView dv,
                         int B, double sigma2, int mean_mode, double tau2, double* loglik,
                         double* beta, int* status) {
  SmallArgs a{};
  a.X = X; a.y = y; a.n = n; a.d = d; a.params = dv.params; a.ldp = dv.ldp; a.K = dv.K;
  a.sigma2 = sigma2; a.mode = mean_mode; a.tau2 = tau2; a.kind = kRowsLoglik; a.mtile = 0;
  a.loglik = loglik; a.beta = beta; a.status = status;
  const int kMaxGrid = 1 << 20;
  for (int b0 = 0; b0 < B; b0 += kMaxGrid) {
    a.draw0 = b0;
    small_launch(s, a, min(kMaxGrid, B - b0), 1);
  }
}

void launch_small_predict(hipStream_t s, const double* X, int n, int d, const double* y, DrawView dv,
                          int S, const double* Xtest, int m, double sigma2, double* mean,
                          double* var, double* beta, int* status) {
  SmallArgs a{};
  a.X = X; a.y = y; a.n = n; a.d = d; a.params = dv.params; a.ldp = dv.ldp; a.K = dv.K;
  a.sigma2 = sigma2; a.mode = 0; a.kind = kRowsPredict; a.Xt = Xtest; a.m = m;
  a.mtile = small_pick_mtile(n, d, m); a.S = S;
  a.mean = mean; a.var = var; a.beta = beta; a.status = status;
  const int nchunks = (m + a.mtile - 1) / a.mtile;
  const int kMaxGrid = 1 << 20;
  for (int b0 = 0; b0 < S; b0 += kMaxGrid) {
    a.draw0 = b0;
    small_launch(s, a, min(kMaxGrid, S - b0), nchunks);
  }
}

void launch_small_inverse(hipStream_t s, const double* X, int n, int d, const double* y, DrawView dv, int draw,
                          double sigma2, double* Rinv, double* loglik, double* beta, int* status) {
  SmallArgs a{};
  // the same elimination that yields R^-1 carries the rows y', 1': likelihood and beta of logpost
  // (HX:454-460) come out of this ONE factorisation (chunk 0 writes them)
  a.X = X; a.y = y; a.n = n; a.d = d; a.params = dv.params; a.ldp = dv.ldp; a.K = dv.K;
  a.sigma2 = sigma2; a.mode = 0; a.kind = kRowsUnit; a.m = n; a.mtile = small_pick_mtile(n, d, n);
  a.draw0 = draw; a.Rinv = Rinv; a.status = status; a.loglik = loglik; a.beta = beta;
  small_launch(s, a, 1, (n + a.mtile - 1) / a.mtile);
}

// gpart must hold B * nchunks * P doubles (nchunks from small_grad_chunks).
int small_grad_chunks(int n, int d) {
  int mt = small_pick_mtile(n, d, n);
  return (n + mt - 1) / mt;
}

void launch_small_grad(hipStream_t s, const double* X, int n, int d, const double* y, DrawView dv,
                       int B, double sigma2, double* loglik, double* beta, double* grad,
                       int* status, double* gpart) {
  SmallArgs a{};
  a.X = X; a.y = y; a.n = n; a.d = d; a.params = dv.params; a.ldp = dv.ldp; a.K = dv.K;
  a.sigma2 = sigma2; a.mode = 0; a.kind = kRowsUnit; a.m = n; a.mtile = small_pick_mtile(n, d, n);
  a.loglik = loglik; a.beta = beta; a.status = status; a.gpart = gpart; a.want_grad = 1;
  const int nchunks = (n + a.mtile - 1) / a.mtile;
  a.draw0 = 0;
  small_launch(s, a, B, nchunks);
  const int P = dv.K + dv.K * d;
  hipLaunchKernelGGL(grad_reduce_kernel, dim3((B * P + 255) / 256), dim3(256), 0, s, gpart, nchunks,
                     P, B, grad);
}

}  // namespace ccgp
