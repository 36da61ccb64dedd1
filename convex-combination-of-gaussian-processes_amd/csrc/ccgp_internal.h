// Internal declarations shared by the HIP translation units of libccgp (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/ccgp.h"

namespace ccgp {

constexpr int kTile = 128;       // block size of the blocked Cholesky (rows/cols per tile)
constexpr int kSmallMaxN = 128;  // n <= this goes to the fused in-LDS evaluator
constexpr int kMaxD = 64;        // input dimensions supported by the covariance kernels
constexpr int kMaxK = 8;         // component GPs per draw
constexpr int kLdsBytes = 160 * 1024;

// one timed launch group: events are recorded on the handle's stream and only read back
// (hipEventElapsedTime) in ccgp_get_timing, so timing never synchronises the pipeline.
struct TimedSpan {
  int id;
  hipEvent_t e0, e1;
};

}  // namespace ccgp

struct ccgp_handle {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  size_t ws_limit = size_t(24) << 30;
  // grow-only device scratch
  void* ws = nullptr;
  size_t ws_bytes = 0;
  // small staging buffers for the host-pointer entry points
  void* stage = nullptr;
  size_t stage_bytes = 0;
  std::string err;
  bool timing = false;
  std::vector<ccgp::TimedSpan> spans;
  size_t spans_used = 0;
};

namespace ccgp {

// Layout of one draw on the device: column-major B x P, element (b, j) at params[b + j*ldp].
struct DrawView {
  const double* params;
  int ldp;  // = B
  int K;
  int d;
};

// ---- cov.hip -------------------------------------------------------------------------
// Dense cross / Gram matrix for ONE draw: out[t + i*ldo], t in [0,m) rows of A (m x d),
// i in [0,n) rows of Bm (n x d).  normalise: divide by sum w^2 (Mixed.corr.*).
void launch_cov_dense(hipStream_t s, const double* A, int m, const double* Bm, int n, int d,
                      DrawView dv, int draw, double* out, int ldo);

// Batched lower-triangle tile writer for the blocked path: for draw b0+z, writes
// s*R_mixed + t into the lower tiles of an npad x npad column-major matrix (identity on
// the padding), z in [0, nb).  scale/shift: mode 0 -> (1, 0); mode 1 -> (sigma2*sum w^2, tau2).
void launch_cov_tiles(hipStream_t s, const double* X, int n, int d, DrawView dv, int b0, int nb,
                      double* Abase, size_t batch_stride, int npad, int mean_mode, double sigma2,
                      double tau2, int ld);
void launch_cov_cross_batched(hipStream_t s, const double* Xtest, int m, const double* X, int n, int d,
                              DrawView dv, int b0, int nb, double* Abase, size_t batch_stride, int ldo);

// ---- small.hip -----------------------------------------------------------------------
// Fused evaluator: one workgroup per draw (and per test-point chunk when m > 0).
size_t small_lds_bytes(int n, int d, int mtile);
int small_pick_mtile(int n, int d, int m);
void launch_small_loglik(hipStream_t s, const double* X, int n, int d, const double* y, DrawView dv,
                         int B, double sigma2, int mean_mode, double tau2, double* loglik,
                         double* beta, int* status);
void launch_small_predict(hipStream_t s, const double* X, int n, int d, const double* y, DrawView dv,
                          int S, const double* Xtest, int m, double sigma2, double* mean,
                          double* var, double* beta, int* status);
// Explicit inverse (solve(R), HX:454) and gradient for small n.
void launch_small_inverse(hipStream_t s, const double* X, int n, int d, DrawView dv, int draw,
                          double* Rinv, int* status);
int small_grad_chunks(int n, int d);
// gpart: scratch of B * small_grad_chunks(n, d) * P doubles
void launch_small_grad(hipStream_t s, const double* X, int n, int d, const double* y, DrawView dv,
                       int B, double sigma2, double* loglik, double* beta, double* grad,
                       int* status, double* gpart);

// ---- small_reg.hip: register-resident evaluator for the plain likelihood (n <= 128) ---------
bool small_reg_supported(int n, int d, bool per_design = false, bool predict = false);
void launch_small_reg_predict(hipStream_t s, const double* X, int n, int d, const double* y, DrawView dv,
                              int S, const double* Xtest, int m, double sigma2, double* mean, double* var,
                              double* beta, int* status);
void launch_small_reg_logdet_designs(hipStream_t s, const double* Xs, int n, int d, DrawView dv, int B,
                                     double* logdet, int* status);
void launch_small_reg_loglik(hipStream_t s, const double* X, int n, int d, const double* y, DrawView dv,
                             int B, double sigma2, int mean_mode, double tau2, double* loglik,
                             double* beta, int* status);

// ---- blocked.hip ---------------------------------------------------------------------
struct BlockedWs {
  double* A;        // nb matrices of ld x npad (lower tiles + right-hand-side / extra tile rows)
  double* invd;     // nb x nt x 128 x 128 inverses of the diagonal blocks
  double* z;        // nb x nt log-det partials
  double* fin;      // nb x 2: s11 = 1'R^-1 1 and beta per matrix (prediction pass)
  size_t a_stride;  // elements between consecutive matrices
  int ld;           // npad + 128 * (1 + ne)
  int ne;           // extra full tile rows (ceil(m / 128) for prediction, else 0)
};
// optional extra work riding on a blocked sweep (n > 128)
enum { kJobPredict = 1, kJobInverse = 2, kJobGrad = 3 };
struct BlockedJob {
  int kind;
  // kJobPredict (a10 + a11): m cross-correlation rows ride along as extra tile rows
  const double* Xtest;  // m x d, device
  int m, S;
  double* mean;         // S x m column-major, device
  double* var;
  // kJobInverse (solve(R), HX:454): identity rows ride along; Rinv = n x n per matrix of the chunk
  double* Rinv;
  // kJobGrad: d loglik / d params; grad is Btot x P column-major
  double* grad;
  int Btot;
  double* gpart;        // scratch: nb x ntiles x P partial sums
  double* alpha;        // scratch: nb x npad,  R^-1 (y - beta 1)
};
bool blocked_grad_supported(int d, int K);
size_t blocked_ws_bytes(int npad, int nb, int ne);
BlockedWs blocked_carve(void* ws, int npad, int nb, int ne);
// factorise nb matrices in place and finish the likelihood; loglik/beta/status are
// indexed from draw b0.
void blocked_loglik(ccgp_handle* h, const double* X, int n, int d, const double* y, DrawView dv,
                    int b0, int nb, int npad, double sigma2, int mean_mode, double tau2,
                    BlockedWs w, double* loglik, double* beta, int* status,
                    const BlockedJob* job = nullptr);

// ---- special.cpp ---------------------------------------------------------------------
void halton_base2(int N, double* out);
double qgamma_unit(double p, double shape);  // quantile of Gamma(shape, rate 1)
double qigamma(double p, double alpha, double beta);

// ---- helpers -------------------------------------------------------------------------
inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Kernel attributes (dynamic-LDS ceiling) are per device: a process that holds handles on several
// GPUs must set them once on EACH.  `mask` is a function-local static, one bit per device.
inline bool first_use_on_device(unsigned long long& mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (mask & bit) return false;
  mask |= bit;
  return true;
}

struct ScopedTimer {
  ccgp_handle* h;
  TimedSpan* sp = nullptr;
  hipStream_t st;
  ScopedTimer(ccgp_handle* hh, int id, hipStream_t stream = nullptr) : h(hh), st(stream ? stream : hh->stream) {
    if (!h->timing) return;
    if (h->spans_used == h->spans.size()) {
      TimedSpan t{};
      (void)hipEventCreate(&t.e0);
      (void)hipEventCreate(&t.e1);
      h->spans.push_back(t);
    }
    sp = &h->spans[h->spans_used++];
    sp->id = id;
    (void)hipEventRecord(sp->e0, st);
  }
  ~ScopedTimer() {
    if (sp) (void)hipEventRecord(sp->e1, st);
  }
};

}  // namespace ccgp
