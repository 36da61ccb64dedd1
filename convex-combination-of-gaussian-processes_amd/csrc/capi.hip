// extern "C" surface of libccgp (include/ccgp.h).  Host orchestration only: argument
// checks, device scratch, chunking of batches, and the handful of host-side scalars
// (log-Jacobian, log-prior, quadrature nodes) that the reference computes in R.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <initializer_list>
#include <map>
#include <new>
#include <vector>

#include "ccgp_internal.h"
#include "special_math.h"

using namespace ccgp;

namespace {

#define CCGP_HIP(call)                                                              \
  do {                                                                              \
    hipError_t e_ = (call);                                                         \
    if (e_ != hipSuccess) {                                                         \
      h->err = std::string(#call) + ": " + hipGetErrorString(e_);                   \
      return CCGP_EHIP;                                                             \
    }                                                                               \
  } while (0)

int fail(ccgp_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  return code;
}

// for the catch handlers: assigning the message may itself throw when memory is gone
int fail_noexcept(ccgp_handle* h, int code, const char* msg) noexcept {
  try {
    if (h) h->err = msg;
  } catch (...) {
  }
  return code;
}

// after enqueueing kernels: launch errors, and a refused kernel-attribute request (raise_lds_limit)
#define CCGP_LAUNCH_CHECK()                                                         \
  do {                                                                              \
    CCGP_HIP(hipGetLastError());                                                    \
    {                                                                               \
      const std::string ae_ = ccgp::attr_error(h->device);                          \
      if (!ae_.empty()) return fail(h, CCGP_EHIP, ae_);                             \
    }                                                                               \
  } while (0)

// the Matern / spline families exist for the 1-D scripts only
int check_family(ccgp_handle* h, const KernelFamily& fam, int d, int K) {
  if (fam.id != 0 && d != 1)
    return fail(h, CCGP_EUNSUPPORTED, "the Matern / spline families are one-dimensional (D1:348-389): d must be 1");
  if (fam.id == 2 && K != 2)
    return fail(h, CCGP_EUNSUPPORTED, "CCGP_KERNEL_MATERN_SPLINE is the two-family script's pair (D1F:453-462): K must be 2");
  return CCGP_OK;
}

int ensure_ws(ccgp_handle* h, size_t bytes) {
  if (bytes <= h->ws_bytes) return CCGP_OK;
  if (h->ws) {
    CCGP_HIP(hipStreamSynchronize(h->stream));
    CCGP_HIP(hipFree(h->ws));
    h->ws = nullptr;
    h->ws_bytes = 0;
  }
  hipError_t e = hipMalloc(&h->ws, bytes);
  if (e != hipSuccess) {
    h->ws = nullptr;
    return fail(h, CCGP_ENOMEM, "device workspace allocation of " + std::to_string(bytes) + " B failed");
  }
  h->ws_bytes = bytes;
  return CCGP_OK;
}

int ensure_stage(ccgp_handle* h, size_t bytes) {
  if (bytes <= h->stage_bytes) return CCGP_OK;
  if (h->stage) {
    CCGP_HIP(hipStreamSynchronize(h->stream));
    CCGP_HIP(hipFree(h->stage));
    h->stage = nullptr;
    h->stage_bytes = 0;
  }
  size_t want = bytes + bytes / 4 + 4096;
  hipError_t e = hipMalloc(&h->stage, want);
  if (e != hipSuccess) {
    h->stage = nullptr;
    return fail(h, CCGP_ENOMEM, "device staging allocation of " + std::to_string(want) + " B failed");
  }
  h->stage_bytes = want;
  return CCGP_OK;
}

// pinned host buffer of the latency path (ccgp_logpost): grow-only
int ensure_pin(ccgp_handle* h, size_t bytes) {
  if (bytes <= h->pin_bytes) return CCGP_OK;
  if (h->pin) {
    CCGP_HIP(hipStreamSynchronize(h->stream));
    (void)hipHostFree(h->pin);
    h->pin = nullptr;
    h->pin_bytes = 0;
  }
  const size_t want = bytes + bytes / 4 + 4096;
  if (hipHostMalloc(&h->pin, want, hipHostMallocDefault) != hipSuccess) {
    h->pin = nullptr;
    return fail(h, CCGP_ENOMEM, "pinned host buffer of " + std::to_string(want) + " B failed");
  }
  h->pin_bytes = want;
  return CCGP_OK;
}

// bump allocator over the staging buffer (256-byte aligned pieces)
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* p) : base(static_cast<char*>(p)) {}
  static size_t al(size_t b) { return (b + 255) / 256 * 256; }
  template <class T>
  T* take(size_t count) {
    T* p = reinterpret_cast<T*>(base + off);
    off += al(count * sizeof(T));
    return p;
  }
};

// ---- host <-> device traffic of the host-pointer entry points ------------------------------------------------
// The pieces of a call lie back to back in the handle's staging buffer (Carver).  A hipMemcpyAsync from / to
// pageable memory is a staged, synchronous copy of its own (10 - 20 us each); what the reference's callers issue
// are many SMALL calls (predict.post per draw and test site HX:688, beta.MLE / factors per draw HX:641), so up to
// kPinMax the host image of the whole span is assembled in the handle's pinned buffer and crosses PCIe in ONE copy
// each way; larger payloads are copied piece by piece as before.
struct Piece {
  void* dev;
  void* host;      // source for push, destination for pull; nullptr: skip
  size_t bytes;
};
constexpr size_t kPinMax = size_t(8) << 20;
constexpr int kPullSlices = ccgp::kPullSlices;

int push(ccgp_handle* h, std::initializer_list<Piece> ps) {
  char *lo = nullptr, *hi = nullptr;
  for (const Piece& p : ps) {
    if (!p.host || !p.bytes) continue;
    char* d = static_cast<char*>(p.dev);
    if (!lo || d < lo) lo = d;
    if (!hi || d + p.bytes > hi) hi = d + p.bytes;
  }
  h->pin_in = 0;
  if (!lo) return CCGP_OK;
  const size_t span = (size_t)(hi - lo);
  if (span <= kPinMax && ensure_pin(h, span) == CCGP_OK) {
    char* pin = static_cast<char*>(h->pin);
    for (const Piece& p : ps)
      if (p.host && p.bytes) std::memcpy(pin + (static_cast<char*>(p.dev) - lo), p.host, p.bytes);
    CCGP_HIP(hipMemcpyAsync(lo, pin, span, hipMemcpyHostToDevice, h->stream));
    h->pin_in = Carver::al(span);
    return CCGP_OK;
  }
  for (const Piece& p : ps)
    if (p.host && p.bytes) CCGP_HIP(hipMemcpyAsync(p.dev, p.host, p.bytes, hipMemcpyHostToDevice, h->stream));
  return CCGP_OK;
}

// device -> host of the result pieces, then the stream is synchronised (the call's results are valid on return)
int pull(ccgp_handle* h, std::initializer_list<Piece> ps) {
  char *lo = nullptr, *hi = nullptr;
  for (const Piece& p : ps) {
    if (!p.host || !p.bytes) continue;
    char* d = static_cast<char*>(p.dev);
    if (!lo || d < lo) lo = d;
    if (!hi || d + p.bytes > hi) hi = d + p.bytes;
  }
  if (!lo) {
    CCGP_HIP(hipStreamSynchronize(h->stream));
    return CCGP_OK;
  }
  const size_t span = (size_t)(hi - lo);
  if (span <= kPinMax && ensure_pin(h, h->pin_in + span) == CCGP_OK) {
    char* pin = static_cast<char*>(h->pin) + h->pin_in;
    // a large result (the S x m tables of ccgp_predict_batch: 2.4 MB per Ground-Vibrations set) comes back in four
    // slices, each followed by an event: the host copies slice c out of the pinned buffer while slice c + 1 is still
    // crossing PCIe, instead of waiting for all of it and then copying all of it
    const int nsl = span >= (size_t(512) << 10) ? kPullSlices : 1;
    if (nsl > 1 && !h->pull_ev[0]) {
      for (int c = 0; c < kPullSlices; ++c)
        if (hipEventCreateWithFlags(&h->pull_ev[c], hipEventDisableTiming) != hipSuccess) {
          for (int q = 0; q < c; ++q) (void)hipEventDestroy(h->pull_ev[q]);
          h->pull_ev[0] = nullptr;
          h->err = "hipEventCreateWithFlags failed";
          return CCGP_EHIP;
        }
    }
    const size_t step = ((span + nsl - 1) / nsl + 255) / 256 * 256;
    for (int c = 0; c < nsl; ++c) {
      const size_t a = std::min(span, (size_t)c * step), b = std::min(span, a + step);
      if (b > a) CCGP_HIP(hipMemcpyAsync(pin + a, lo + a, b - a, hipMemcpyDeviceToHost, h->stream));
      if (nsl > 1) CCGP_HIP(hipEventRecord(h->pull_ev[c], h->stream));
    }
    for (int c = 0; c < nsl; ++c) {
      const size_t a = std::min(span, (size_t)c * step), b = std::min(span, a + step);
      if (nsl > 1) CCGP_HIP(hipEventSynchronize(h->pull_ev[c]));
      else CCGP_HIP(hipStreamSynchronize(h->stream));
      for (const Piece& p : ps) {
        if (!p.host || !p.bytes) continue;
        const size_t p0 = (size_t)(static_cast<char*>(p.dev) - lo), p1 = p0 + p.bytes;
        const size_t x0 = std::max(p0, a), x1 = std::min(p1, b);
        if (x1 > x0) std::memcpy(static_cast<char*>(p.host) + (x0 - p0), pin + x0, x1 - x0);
      }
    }
    return CCGP_OK;
  }
  for (const Piece& p : ps)
    if (p.host && p.bytes) CCGP_HIP(hipMemcpyAsync(p.host, p.dev, p.bytes, hipMemcpyDeviceToHost, h->stream));
  CCGP_HIP(hipStreamSynchronize(h->stream));
  return CCGP_OK;
}

template <class T>
Piece piece(T* dev, const T* host, size_t count) {
  return Piece{dev, const_cast<T*>(host), sizeof(T) * count};
}

bool bad_shape(int n, int d, int K) {
  return n < 1 || d < 1 || d > kMaxD || K < 1 || K > kMaxK;
}

// blocked-path chunk size (matrices per pass) under the workspace limit
int blocked_chunk(const ccgp_handle* h, int npad, int B, int ne = 0) {
  size_t per = blocked_ws_bytes(npad, 1, ne);
  // never plan beyond what the device can give right now (other handles / processes may share it):
  // free memory plus what this handle would release by regrowing, less a margin
  size_t limit = h->ws_limit, free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
    const size_t margin = size_t(1) << 30;
    const size_t avail = free_b + h->ws_bytes > margin ? free_b + h->ws_bytes - margin : 0;
    if (avail < limit) limit = avail;
  }
  size_t nb = limit / per;
  if (nb < 1) nb = 1;
  if (nb > (size_t)B) nb = B;
  if (nb > 65535) nb = 65535;   // the chunk index is a grid y / z dimension in cov_kernel, rhs_rows_kernel, ...
  return (int)nb;
}

int loglik_dev(ccgp_handle* h, const double* dX, int n, int d, const double* dy, int K,
               const double* dparams, int B, double sigma2, int mean_mode, double tau2,
               double* d_loglik, double* d_beta, int* d_status) {
  if (bad_shape(n, d, K) || B < 0 || !dX || !dy || !dparams || !d_loglik || !d_status)
    return fail(h, CCGP_EINVAL, "ccgp_loglik_batch: bad argument");
  if (mean_mode != CCGP_MEAN_PROFILE_BETA && mean_mode != CCGP_MEAN_ZERO_PLUS_TAU2)
    return fail(h, CCGP_EINVAL, "ccgp_loglik_batch: unknown mean_mode");
  if (B == 0) return CCGP_OK;
  DrawView dv{dparams, B, K, d};
  dv.fam = h->fam;
  if (int frc = check_family(h, dv.fam, d, K)) return frc;
  // the fused evaluators generate Gaussian correlations in registers; any other family goes through
  // the materialised-matrix (blocked) path, where only cov_kernel knows about families
  const bool gauss = dv.fam.id == 0;
  const bool reg_ok = gauss && small_reg_supported(n, d, K);
  const bool lds_ok = gauss && n <= kSmallMaxN && small_lds_bytes(n, d, 0) <= (size_t)kLdsBytes - 64;
  if (reg_ok || lds_ok) {   // otherwise (n > 128, or d too large for LDS) the blocked path takes it
    ScopedTimer t(h, CCGP_T_FUSED);
    if (reg_ok)
      launch_small_reg_loglik(h->stream, dX, n, d, dy, dv, B, sigma2, mean_mode, tau2, d_loglik, d_beta,
                              d_status, h->opt_small_grid16 != 0);
    else
      launch_small_loglik(h->stream, dX, n, d, dy, dv, B, sigma2, mean_mode, tau2, d_loglik, d_beta,
                          d_status);
    CCGP_LAUNCH_CHECK();
    return CCGP_OK;
  }
  const int npad = round_up(n, kTile);
  int nbc = blocked_chunk(h, npad, B);
  int rc = ensure_ws(h, blocked_ws_bytes(npad, nbc, 0));
  while (rc == CCGP_ENOMEM && nbc > 1) {   // another handle / process took the memory in between: smaller chunks
    nbc = (nbc + 1) / 2;
    rc = ensure_ws(h, blocked_ws_bytes(npad, nbc, 0));
  }
  if (rc) return rc;
  CCGP_HIP(hipMemsetAsync(d_status, 0, sizeof(int) * (size_t)B, h->stream));
  for (int b0 = 0; b0 < B; b0 += nbc) {
    const int nb = std::min(nbc, B - b0);
    BlockedWs w = blocked_carve(h->ws, npad, nb, 0);
    blocked_loglik(h, dX, n, d, dy, dv, b0, nb, npad, sigma2, mean_mode, tau2, w, d_loglik, d_beta,
                   d_status);
  }
  CCGP_LAUNCH_CHECK();
  return CCGP_OK;
}

int count_bad(const int* status, int B) {
  int c = 0;
  for (int i = 0; i < B; ++i) c += status[i] != 0;
  return c;
}

// ---- tiny kernels for the literal R.Inv-based helpers (a6, a7, a10, a11) -----------------
// (Rounds 1 - 3 let one thread walk a whole row or column of R.Inv alone: ~20 us per kernel for 4096 multiply-adds at
// n = 64, a third of a literal call.  Now lane = row -- a column of R.Inv is contiguous --, the four waves share the columns
// and combine in LDS in fixed order.)
__device__ inline double wg4_sum(double x, double (&red)[4], int tid) {
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = x;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void rinv_terms_kernel(const double* Rinv, const double* y, int n, double beta,
                                                         double* mean_factor, double* colsum, double* scal) {
  // one workgroup; scal[0] = 1'Rinv y, scal[1] = sum(Rinv), scal[2] = (y-b)'Rinv(y-b).
  // The caller's R.Inv is taken as it is, symmetric or not (R's solve() output is symmetric only up to rounding):
  // var.factor1 = apply(R.Inv, 2, sum) are COLUMN sums (HX:609), 1'R.Inv y = sum_j colsum_j y_j (HX:387), and
  // mean.factor = R.Inv %*% (y - beta) are row dot products (HX:608).
  __shared__ double part[4][64];
  __shared__ double red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  for (int i0 = 0; i0 < n; i0 += 64) {
    // rows i0 .. i0 + 63: (R.Inv (y - beta))_i, lane = row (a column of R.Inv is contiguous), the waves share the columns
    const int i = i0 + lane;
    double mf = 0.0;
    if (i < n) {
#pragma unroll 4
      for (int j = wave; j < n; j += 4) mf = fma(Rinv[i + (size_t)j * n], y[j] - beta, mf);
    }
    part[wave][lane] = mf;
    __syncthreads();
    if (wave == 0 && i < n) {
      mf = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
      if (mean_factor) mean_factor[i] = mf;
      a2 += (y[i] - beta) * mf;
    }
    __syncthreads();
  }
  // column sums: four lanes per column (rows i = q, q + 4, ...), combined in fixed order
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int j = j0 + (tid >> 2), q = tid & 3;
    double cs = 0.0;
    if (j < n) {
      const double* col = Rinv + (size_t)j * n;
#pragma unroll 4
      for (int i = q; i < n; i += 4) cs += col[i];
    }
    cs += __shfl_xor(cs, 1, 64);
    cs += __shfl_xor(cs, 2, 64);
    if (j < n && q == 0) {
      if (colsum) colsum[j] = cs;
      a0 += cs * y[j];
      a1 += cs;
    }
  }
  a0 = wg4_sum(a0, red, tid);
  a1 = wg4_sum(a1, red, tid);
  a2 = wg4_sum(a2, red, tid);
  if (tid == 0) { scal[0] = a0; scal[1] = a1; scal[2] = a2; }
}

__global__ __launch_bounds__(256) void predict_factors_kernel(const double* r, int m, int n, double beta,
                                                              const double* mean_factor, const double* v1, double v2,
                                                              const double* Rinv, double sigma2, double* mean, double* var) {
  // one workgroup per test point t; r is m x n column-major (r[t + i*m])
  __shared__ double part[4][64];
  __shared__ double red[4];
  const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double q = 0.0, s1 = 0.0, sm = 0.0;
  for (int i0 = 0; i0 < n; i0 += 64) {
    const int i = i0 + lane;
    double acc = 0.0;
    if (i < n) {
#pragma unroll 4
      for (int j = wave; j < n; j += 4) acc = fma(Rinv[i + (size_t)j * n], r[t + (size_t)j * m], acc);
    }
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && i < n) {
      acc = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
      const double ri = r[t + (size_t)i * m];
      q = fma(ri, acc, q);
      s1 = fma(v1[i], ri, s1);
      sm = fma(mean_factor[i], ri, sm);
    }
    __syncthreads();
  }
  q = wg4_sum(q, red, tid);
  s1 = wg4_sum(s1, red, tid);
  sm = wg4_sum(sm, red, tid);
  if (tid == 0) {
    const double u = 1.0 - s1;
    var[t] = sigma2 * (1.0 - q + u * u / v2);
    mean[t] = beta + sm;
  }
}

// log-mean-exp of each grid row's N conditional log-likelihoods (likeli.hyperpars, HX:574):
// logs is laid out [g*N + j]; NaN entries (non-PD draws) propagate as in R's mean().
__global__ void row_logmeanexp_kernel(const double* logs, int N, int take_log, double* out) {
  __shared__ double red[4];
  const int g = blockIdx.x, tid = threadIdx.x;
  const double* row = logs + (size_t)g * N;
  double mx = -INFINITY;
  bool nan = false;
  for (int j = tid; j < N; j += blockDim.x) {
    double v = row[j];
    if (v != v) nan = true;
    mx = fmax(mx, v);
  }
  for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_down(mx, off, 64));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  __syncthreads();
  double s = 0.0;
  for (int j = tid; j < N; j += blockDim.x) s += exp(row[j] - mx);
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  int anynan = __syncthreads_or(nan ? 1 : 0);
  if (tid == 0) {
    double tot = red[0] + red[1] + red[2] + red[3];
    double lme = mx + log(tot / N);
    double v = take_log ? lme : exp(lme);
    if (anynan) v = __longlong_as_double(0x7ff8000000000000LL);
    out[g] = v;
  }
}

// ---- hyperprior grid: the G x N table of draws, built on the device (likeli.hyperpars HX:554-559) ----------
// qtab[s * N + j] = qgamma(1 - u_j, shape_s, rate 1),  u_j = runif.halton(N, 1)[j]
__global__ void grid_qtab_kernel(const double* shapes, int ns, int N, double* qtab) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)ns * N) return;
  const int s = (int)(idx / N), j = (int)(idx % N);
  qtab[idx] = qgamma_unit(1.0 - halton2((unsigned)j + 1u), shapes[s]);
}

// draw b = g N + j: p = u_j, theta1 = qigamma(u_j, a1, b1) = b1 / qtab[a1][j], theta2 likewise (HX:554-556);
// isotropic: theta_c repeated over the d dimensions; anisotropic (ANI:399-406 with the grid's quantiles per
// dimension, BASELINE config 3): component 1 = (theta1, theta2), component 2 = (1 + lambda) (theta1, theta2)
__global__ void grid_expand_kernel(const double* hyper, const int* shape_idx, const double* qtab, int G, int N,
                                   int d, int aniso, double lambda, double* params) {
  const size_t B = (size_t)G * N;
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int g = (int)(b / N), j = (int)(b % N);
  const double p = halton2((unsigned)j + 1u);
  const double th1 = hyper[g + (size_t)G] / qtab[(size_t)shape_idx[g] * N + j];
  const double th2 = hyper[g + (size_t)3 * G] / qtab[(size_t)shape_idx[G + g] * N + j];
  params[b] = p;
  params[b + B] = 1.0 - p;
  if (aniso) {
    params[b + 2 * B] = th1;
    params[b + 3 * B] = th2;
    params[b + 4 * B] = (1.0 + lambda) * th1;
    params[b + 5 * B] = (1.0 + lambda) * th2;
  } else {
    for (int k = 0; k < d; ++k) {
      params[b + (size_t)(2 + k) * B] = th1;
      params[b + (size_t)(2 + d + k) * B] = th2;
    }
  }
}

// number of evaluations whose factorisation met a non-positive pivot (the C ABI's positive return value)
__global__ void count_bad_kernel(const int* status, int B, int* out) {
  int c = 0;
  const int end = min(B, (int)(blockIdx.x + 1) * 1024);
  for (int i = blockIdx.x * 1024 + threadIdx.x; i < end; i += 256) c += status[i] != 0;
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

}  // namespace

// No C++ exception may cross the C ABI (inside R that would be std::terminate for the user's session): every entry point
// that takes a handle is a function-try-block; host-memory exhaustion (std::vector / std::string growth) comes back as
// CCGP_ENOMEM, anything else as CCGP_EHIP, with the message in ccgp_last_error.
#define CCGP_GUARD_END(h)                                                                    \
  catch (const std::bad_alloc&) {                                                            \
    return fail_noexcept(h, CCGP_ENOMEM, "host memory exhausted inside libccgp");            \
  } catch (const std::exception& e_) {                                                       \
    return fail_noexcept(h, CCGP_EHIP, e_.what());                                           \
  } catch (...) {                                                                            \
    return fail_noexcept(h, CCGP_EHIP, "unknown C++ exception inside libccgp");              \
  }

extern "C" {

const char* ccgp_version(void) { return "ccgp-mi355x 0.1 (gfx950)"; }

int ccgp_create(int device, ccgp_handle** out) {
  if (!out) return CCGP_EINVAL;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return CCGP_EHIP;
  if (hipSetDevice(device) != hipSuccess) return CCGP_EHIP;
  ccgp_handle* h = new (std::nothrow) ccgp_handle();
  if (!h) return CCGP_ENOMEM;
  h->device = device;
  if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete h;
    return CCGP_EHIP;
  }
  h->stream = h->own_stream;
  // default scratch cap: three quarters of the device (216 of 288 GB on MI355X -- the whole 512-point
  // n = 4096 grid of BASELINE config 4 is 73 GB and runs as ONE chunk)
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) h->ws_limit = total_b / 4 * 3;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) h->n_cus = cus;
  if (const char* e = std::getenv("CCGP_SCHED_BACKLOG")) h->sched_backlog_min = std::atoi(e);
  if (const char* e = std::getenv("CCGP_SCHED_TIMEOUT_MS")) {
    const int v = std::atoi(e);
    if (v > 0) h->sched_timeout_ms = v;
  }
  *out = h;
  return CCGP_OK;
}

int ccgp_destroy(ccgp_handle* h) try {
  if (!h) return CCGP_OK;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  for (auto& s : h->spans) {
    (void)hipEventDestroy(s.e0);
    (void)hipEventDestroy(s.e1);
  }
  if (h->ws) (void)hipFree(h->ws);
  if (h->stage) (void)hipFree(h->stage);
  if (h->pin) (void)hipHostFree(h->pin);
  for (auto& e : h->pull_ev)
    if (e) (void)hipEventDestroy(e);
  if (h->aux_fork) (void)hipEventDestroy(h->aux_fork);
  if (h->aux_join) (void)hipEventDestroy(h->aux_join);
  if (h->aux_stream) (void)hipStreamDestroy(h->aux_stream);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
  return CCGP_OK;
} CCGP_GUARD_END(h)

const char* ccgp_last_error(const ccgp_handle* h) { return h ? h->err.c_str() : "null handle"; }

int ccgp_set_stream(ccgp_handle* h, void* hip_stream) try {
  if (!h) return CCGP_EINVAL;
  h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
  return CCGP_OK;
} CCGP_GUARD_END(h)

int ccgp_set_kernel(ccgp_handle* h, int family, double nu) try {
  if (!h) return CCGP_EINVAL;
  if (family == CCGP_KERNEL_GAUSS) {
    h->fam = ccgp::KernelFamily{};
    return CCGP_OK;
  }
  if ((family != CCGP_KERNEL_MATERN && family != CCGP_KERNEL_MATERN_SPLINE) || !(nu > 1.0) || !(nu <= 10.0))
    return fail(h, CCGP_EINVAL, "ccgp_set_kernel: family must be CCGP_KERNEL_GAUSS, or CCGP_KERNEL_MATERN / CCGP_KERNEL_MATERN_SPLINE with 1 < nu <= 10 (the range the quadrature is validated on; the scripts use 5)");
  h->fam.id = family;
  h->fam.nu = nu;
  h->fam.norm = 1.0 / (std::tgamma(nu) * std::pow(2.0, nu - 1.0));
  return CCGP_OK;
} CCGP_GUARD_END(h)

int ccgp_set_workspace_limit(ccgp_handle* h, size_t bytes) try {
  if (!h || bytes < (size_t(1) << 20)) return CCGP_EINVAL;
  h->ws_limit = bytes;
  return CCGP_OK;
} CCGP_GUARD_END(h)

int ccgp_set_option(ccgp_handle* h, int option, int value) try {
  if (!h) return CCGP_EINVAL;
  if (option == CCGP_OPT_TAIL_STRIPS && value >= 0 && value <= 2) {
    h->opt_tail_strips = value;
    return CCGP_OK;
  }
  if (option == CCGP_OPT_FUSE_DIAG && (value == 0 || value == 1)) {
    h->opt_fuse_diag = value;
    return CCGP_OK;
  }
  if (option == CCGP_OPT_WIDE_OFFSETS && (value == 0 || value == 1)) {
    h->opt_wide_offsets = value;
    return CCGP_OK;
  }
  if (option == CCGP_OPT_SMALL_GRID16 && (value == 0 || value == 1)) {
    h->opt_small_grid16 = value;
    return CCGP_OK;
  }
  if (option == CCGP_OPT_PREDICT_FACTOR && (value == 0 || value == 1)) {
    h->opt_predict_factor = value;
    return CCGP_OK;
  }
  if (option == CCGP_OPT_SCHED && value >= 0 && value <= 3) {
    h->opt_sched = value;
    return CCGP_OK;
  }
  if (option == CCGP_OPT_SCHED_POLICY && value >= 0 && value <= 31) {
    h->opt_sched_policy = value;
    return CCGP_OK;
  }
  return fail(h, CCGP_EINVAL, "ccgp_set_option: unknown option or value");
} CCGP_GUARD_END(h)

int ccgp_synchronize(ccgp_handle* h) try {
  if (!h) return CCGP_EINVAL;
  CCGP_HIP(hipStreamSynchronize(h->stream));
  return CCGP_OK;
} CCGP_GUARD_END(h)

int ccgp_reserve(ccgp_handle* h, int n, int d, int K, int B, int m) try {
  if (!h || bad_shape(n, d, K) || B < 1 || m < 0) return fail(h, CCGP_EINVAL, "ccgp_reserve: bad argument");
  CCGP_HIP(hipSetDevice(h->device));
  if (n > kSmallMaxN || h->fam.id != 0) {
    const int npad = round_up(n, kTile);
    const int ne = (m + kTile - 1) / kTile;
    int rc = ensure_ws(h, blocked_ws_bytes(npad, blocked_chunk(h, npad, B, ne), ne));
    if (rc) return rc;
  }
  const int P = K + K * d;
  size_t st = Carver::al(sizeof(double) * (size_t)n * d) + Carver::al(sizeof(double) * n) +
              Carver::al(sizeof(double) * (size_t)B * P) + 3 * Carver::al(sizeof(double) * B) +
              Carver::al(sizeof(double) * (size_t)m * d) + 2 * Carver::al(sizeof(double) * (size_t)B * m) + 4096;
  return ensure_stage(h, st);
} CCGP_GUARD_END(h)

int ccgp_enable_timing(ccgp_handle* h, int on) try {
  if (!h) return CCGP_EINVAL;
  // on = 1: every launch group; otherwise a bit mask, bit (1 + id) selects CCGP_T_<id> (bench.py times only
  // the update launches inside its timed region: two event records per launch are not free)
  h->timing = on == 1 ? ~0u : (unsigned)on >> 1;
  h->spans_used = 0;
  return CCGP_OK;
} CCGP_GUARD_END(h)

int ccgp_get_timing(ccgp_handle* h, int id, double* out_ms, int* out_launches) try {
  if (!h || id < 0 || id >= CCGP_T_COUNT) return CCGP_EINVAL;
  CCGP_HIP(hipStreamSynchronize(h->stream));
  double ms = 0.0;
  int cnt = 0;
  for (size_t i = 0; i < h->spans_used; ++i) {
    if (h->spans[i].id != id) continue;
    float f = 0.f;
    CCGP_HIP(hipEventElapsedTime(&f, h->spans[i].e0, h->spans[i].e1));
    ms += f;
    ++cnt;
  }
  if (out_ms) *out_ms = ms;
  if (out_launches) *out_launches = cnt;
  return CCGP_OK;
} CCGP_GUARD_END(h)

int ccgp_last_sched_profile(ccgp_handle* h, unsigned long long* out, int max_workgroups, int* out_workgroups) try {
  if (!h || !out || max_workgroups < 1) return CCGP_EINVAL;
  CCGP_HIP(hipSetDevice(h->device));
  CCGP_HIP(hipStreamSynchronize(h->stream));
  const int nw = std::min(max_workgroups, h->sched_prof_wgs);
  if (out_workgroups) *out_workgroups = nw;
  if (nw > 0 && h->sched_prof_dev)
    CCGP_HIP(hipMemcpy(out, h->sched_prof_dev, sizeof(unsigned long long) * 8 * (size_t)nw, hipMemcpyDeviceToHost));
  return CCGP_OK;
} CCGP_GUARD_END(h)

// ---- a1-a5 ------------------------------------------------------------------------------
static int corr_common(ccgp_handle* h, const double* Xnew, int m, const double* X, int n, int d,
                       int K, const double* params_row, double* out, bool gram) {
  if (!h || bad_shape(n, d, K) || m < 1 || !X || !params_row || !out || (!gram && !Xnew))
    return fail(h, CCGP_EINVAL, "ccgp_corr_*: bad argument");
  CCGP_HIP(hipSetDevice(h->device));
  const int P = K + K * d;
  size_t need = Carver::al(sizeof(double) * (size_t)n * d) + Carver::al(sizeof(double) * (size_t)m * d) +
                Carver::al(sizeof(double) * P) + Carver::al(sizeof(double) * (size_t)m * n);
  int rc = ensure_stage(h, need);
  if (rc) return rc;
  Carver c(h->stage);
  double* dX = c.take<double>((size_t)n * d);
  double* dXn = c.take<double>((size_t)m * d);
  double* dp = c.take<double>(P);
  double* dout = c.take<double>((size_t)m * n);
  if (int prc = push(h, {piece(dX, X, (size_t)n * d), piece(dXn, gram ? nullptr : Xnew, (size_t)m * d), piece(dp, params_row, P)}))
    return prc;
  DrawView dv{dp, 1, K, d};
  dv.fam = h->fam;
  if (int frc = check_family(h, dv.fam, d, K)) return frc;
  {
    ScopedTimer t(h, CCGP_T_COV);
    launch_cov_dense(h->stream, gram ? dX : dXn, m, dX, n, d, dv, 0, dout, m);
  }
  CCGP_LAUNCH_CHECK();
  return pull(h, {piece(dout, out, (size_t)m * n)});
}

int ccgp_corr_matrix(ccgp_handle* h, const double* X, int n, int d, const double* theta,
                     double* out_R) try {
  if (!theta || d < 1 || d > kMaxD) return fail(h, CCGP_EINVAL, "ccgp_corr_matrix: bad argument");
  std::vector<double> row(1 + d);
  row[0] = 1.0;
  for (int k = 0; k < d; ++k) row[1 + k] = theta[k];
  return corr_common(h, nullptr, n, X, n, d, 1, row.data(), out_R, true);
} CCGP_GUARD_END(h)

int ccgp_corr_cross(ccgp_handle* h, const double* Xnew, int m, const double* X, int n, int d,
                    const double* theta, double* out) try {
  if (!theta || d < 1 || d > kMaxD) return fail(h, CCGP_EINVAL, "ccgp_corr_cross: bad argument");
  std::vector<double> row(1 + d);
  row[0] = 1.0;
  for (int k = 0; k < d; ++k) row[1 + k] = theta[k];
  return corr_common(h, Xnew, m, X, n, d, 1, row.data(), out, false);
} CCGP_GUARD_END(h)

int ccgp_mixed_corr_matrix(ccgp_handle* h, const double* X, int n, int d, int K,
                           const double* params, double* out_R) try {
  return corr_common(h, nullptr, n, X, n, d, K, params, out_R, true);
} CCGP_GUARD_END(h)

int ccgp_mixed_corr_cross(ccgp_handle* h, const double* Xnew, int m, const double* X, int n,
                          int d, int K, const double* params, double* out) try {
  return corr_common(h, Xnew, m, X, n, d, K, params, out, false);
} CCGP_GUARD_END(h)

// ---- a6, a7, a10 ---------------------------------------------------------------------------
static int rinv_terms(ccgp_handle* h, const double* R_inv, const double* y, int n, double beta,
                      double* mean_factor, double* colsum, double scal[3]) {
  if (!h || n < 1 || !R_inv || !y) return fail(h, CCGP_EINVAL, "R.Inv helper: bad argument");
  CCGP_HIP(hipSetDevice(h->device));
  size_t need = Carver::al(sizeof(double) * (size_t)n * n) + 3 * Carver::al(sizeof(double) * n) + 256;
  int rc = ensure_stage(h, need);
  if (rc) return rc;
  Carver c(h->stage);
  double* dR = c.take<double>((size_t)n * n);
  double* dy = c.take<double>(n);
  double* dmf = c.take<double>(n);
  double* dcs = c.take<double>(n);
  double* dsc = c.take<double>(4);
  if (int prc = push(h, {piece(dR, R_inv, (size_t)n * n), piece(dy, y, n)})) return prc;
  hipLaunchKernelGGL(rinv_terms_kernel, dim3(1), dim3(256), 0, h->stream, dR, dy, n, beta, dmf, dcs, dsc);
  CCGP_LAUNCH_CHECK();
  return pull(h, {piece(dmf, mean_factor, n), piece(dcs, colsum, n), piece(dsc, scal, 3)});
}

int ccgp_beta_mle(ccgp_handle* h, const double* R_inv, const double* y, int n, double* out_beta) try {
  if (!out_beta) return fail(h, CCGP_EINVAL, "ccgp_beta_mle: bad argument");
  double sc[3];
  int rc = rinv_terms(h, R_inv, y, n, 0.0, nullptr, nullptr, sc);
  if (rc) return rc;
  *out_beta = sc[0] / sc[1];
  return CCGP_OK;
} CCGP_GUARD_END(h)

int ccgp_sigma2_mle(ccgp_handle* h, const double* R_inv, const double* y, int n, double beta,
                    double* out_sigma2) try {
  if (!out_sigma2) return fail(h, CCGP_EINVAL, "ccgp_sigma2_mle: bad argument");
  double sc[3];
  int rc = rinv_terms(h, R_inv, y, n, beta, nullptr, nullptr, sc);
  if (rc) return rc;
  *out_sigma2 = sc[2] / n;
  return CCGP_OK;
} CCGP_GUARD_END(h)

int ccgp_factors(ccgp_handle* h, const double* R_inv, double beta, const double* y, int n,
                 double* out) try {
  if (!out) return fail(h, CCGP_EINVAL, "ccgp_factors: bad argument");
  double sc[3];
  int rc = rinv_terms(h, R_inv, y, n, beta, out, out + n, sc);
  if (rc) return rc;
  out[2 * n] = sc[1];
  return CCGP_OK;
} CCGP_GUARD_END(h)

int ccgp_predict_from_factors(ccgp_handle* h, const double* r, int m, int n, double beta,
                              const double* mean_factor, const double* var_factor1,
                              double var_factor2, const double* R_inv, double sigma2,
                              double* out_mean, double* out_var) try {
  if (!h || m < 1 || n < 1 || !r || !mean_factor || !var_factor1 || !R_inv || !out_mean || !out_var)
    return fail(h, CCGP_EINVAL, "ccgp_predict_from_factors: bad argument");
  CCGP_HIP(hipSetDevice(h->device));
  size_t need = Carver::al(sizeof(double) * (size_t)n * n) + Carver::al(sizeof(double) * (size_t)m * n) +
                2 * Carver::al(sizeof(double) * n) + 2 * Carver::al(sizeof(double) * m);
  int rc = ensure_stage(h, need);
  if (rc) return rc;
  Carver c(h->stage);
  double* dR = c.take<double>((size_t)n * n);
  double* dr = c.take<double>((size_t)m * n);
  double* dmf = c.take<double>(n);
  double* dv1 = c.take<double>(n);
  double* dmean = c.take<double>(m);
  double* dvar = c.take<double>(m);
  if (int prc = push(h, {piece(dR, R_inv, (size_t)n * n), piece(dr, r, (size_t)m * n), piece(dmf, mean_factor, n),
                         piece(dv1, var_factor1, n)}))
    return prc;
  hipLaunchKernelGGL(predict_factors_kernel, dim3(m), dim3(256), 0, h->stream, dr, m, n, beta, dmf,
                     dv1, var_factor2, dR, sigma2, dmean, dvar);
  CCGP_LAUNCH_CHECK();
  return pull(h, {piece(dmean, out_mean, m), piece(dvar, out_var, m)});
} CCGP_GUARD_END(h)

int ccgp_predict_post(ccgp_handle* h, const double* Xnew, int m, const double* X, int n, int d, int K,
                      const double* params_row, double beta, const double* mean_factor,
                      const double* var_factor1, double var_factor2, const double* R_inv, double sigma2,
                      double* out_mean, double* out_var) try {
  if (!h || bad_shape(n, d, K) || m < 1 || !Xnew || !X || !params_row || !mean_factor || !var_factor1 || !R_inv ||
      !out_mean || !out_var)
    return fail(h, CCGP_EINVAL, "ccgp_predict_post: bad argument");
  CCGP_HIP(hipSetDevice(h->device));
  const int P = K + K * d;
  size_t need = Carver::al(sizeof(double) * (size_t)n * d) + Carver::al(sizeof(double) * (size_t)m * d) +
                Carver::al(sizeof(double) * P) + 2 * Carver::al(sizeof(double) * n) +
                Carver::al(sizeof(double) * (size_t)n * n) + Carver::al(sizeof(double) * (size_t)m * n) +
                2 * Carver::al(sizeof(double) * m);
  int rc = ensure_stage(h, need);
  if (rc) return rc;
  Carver c(h->stage);
  double* dX = c.take<double>((size_t)n * d);
  double* dXn = c.take<double>((size_t)m * d);
  double* dp = c.take<double>(P);
  double* dmf = c.take<double>(n);
  double* dv1 = c.take<double>(n);
  double* dR = c.take<double>((size_t)n * n);
  double* dr = c.take<double>((size_t)m * n);
  double* dmean = c.take<double>(m);
  double* dvar = c.take<double>(m);
  if (int prc = push(h, {piece(dX, X, (size_t)n * d), piece(dXn, Xnew, (size_t)m * d), piece(dp, params_row, P),
                         piece(dmf, mean_factor, n), piece(dv1, var_factor1, n), piece(dR, R_inv, (size_t)n * n)}))
    return prc;
  DrawView dv{dp, 1, K, d};
  dv.fam = h->fam;
  if (int frc = check_family(h, dv.fam, d, K)) return frc;
  {
    ScopedTimer t(h, CCGP_T_COV);   // r = Mixed.corr.vec(x_t, D.train, ...) (HX:665), exactly ccgp_mixed_corr_cross's kernel
    launch_cov_dense(h->stream, dXn, m, dX, n, d, dv, 0, dr, m);
  }
  hipLaunchKernelGGL(predict_factors_kernel, dim3(m), dim3(256), 0, h->stream, dr, m, n, beta, dmf, dv1,
                     var_factor2, dR, sigma2, dmean, dvar);
  CCGP_LAUNCH_CHECK();
  return pull(h, {piece(dmean, out_mean, m), piece(dvar, out_var, m)});
} CCGP_GUARD_END(h)

// ---- a8/a9/a12 -------------------------------------------------------------------------------
int ccgp_loglik_batch_dev(ccgp_handle* h, const double* dX, int n, int d, const double* dy, int K,
                          const double* dparams, int B, double sigma2, int mean_mode, double tau2,
                          double* d_loglik, double* d_beta, int* d_status) try {
  if (!h) return CCGP_EINVAL;
  CCGP_HIP(hipSetDevice(h->device));
  return loglik_dev(h, dX, n, d, dy, K, dparams, B, sigma2, mean_mode, tau2, d_loglik, d_beta,
                    d_status);
} CCGP_GUARD_END(h)

int ccgp_loglik_batch(ccgp_handle* h, const double* X, int n, int d, const double* y, int K,
                      const double* params, int B, double sigma2, int mean_mode, double tau2,
                      double* out_loglik, double* out_beta, int* status) try {
  if (!h) return CCGP_EINVAL;
  if (bad_shape(n, d, K) || B < 0 || !X || !y || !params || !out_loglik)
    return fail(h, CCGP_EINVAL, "ccgp_loglik_batch: bad argument");
  if (B == 0) return CCGP_OK;
  CCGP_HIP(hipSetDevice(h->device));
  const int P = K + K * d;
  // inputs X | y | params and results loglik | beta | status, each contiguous on the device
  const size_t in_d = (size_t)n * d + n + (size_t)B * P;
  const size_t out_d = 2 * (size_t)B + ((size_t)B + 1) / 2;
  int rc = ensure_stage(h, Carver::al(sizeof(double) * in_d) + Carver::al(sizeof(double) * out_d));
  if (rc) return rc;
  Carver c(h->stage);
  double* din = c.take<double>(in_d);
  double* dout = c.take<double>(out_d);
  double* dX = din;
  double* dy = din + (size_t)n * d;
  double* dp = dy + n;
  double* dll = dout;
  double* dbeta = dout + B;
  int* dst = reinterpret_cast<int*>(dout + 2 * (size_t)B);
  // A small call (a speculative batch of Metropolis candidates, the points of a numerical derivative) is dominated by
  // the host side: through pageable memory every one of the three uploads and three downloads is a staged,
  // synchronous copy of its own.  Up to 1 MiB the payload goes through the handle's pinned buffer: ONE copy each way.
  const bool pinned = sizeof(double) * (in_d + out_d) <= (size_t(1) << 20) && ensure_pin(h, sizeof(double) * (in_d + out_d)) == CCGP_OK;
  if (pinned) {
    double* pin = static_cast<double*>(h->pin);
    std::memcpy(pin, X, sizeof(double) * (size_t)n * d);
    std::memcpy(pin + (size_t)n * d, y, sizeof(double) * n);
    std::memcpy(pin + (size_t)n * d + n, params, sizeof(double) * (size_t)B * P);
    CCGP_HIP(hipMemcpyAsync(din, pin, sizeof(double) * in_d, hipMemcpyHostToDevice, h->stream));
  } else {
    CCGP_HIP(hipMemcpyAsync(dX, X, sizeof(double) * (size_t)n * d, hipMemcpyHostToDevice, h->stream));
    CCGP_HIP(hipMemcpyAsync(dy, y, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
    CCGP_HIP(hipMemcpyAsync(dp, params, sizeof(double) * (size_t)B * P, hipMemcpyHostToDevice, h->stream));
  }
  rc = loglik_dev(h, dX, n, d, dy, K, dp, B, sigma2, mean_mode, tau2, dll, dbeta, dst);
  if (rc) return rc;
  std::vector<int> st(B);
  if (pinned) {
    double* pout = static_cast<double*>(h->pin) + in_d;
    CCGP_HIP(hipMemcpyAsync(pout, dout, sizeof(double) * out_d, hipMemcpyDeviceToHost, h->stream));
    CCGP_HIP(hipStreamSynchronize(h->stream));
    std::memcpy(out_loglik, pout, sizeof(double) * B);
    if (out_beta) std::memcpy(out_beta, pout + B, sizeof(double) * B);
    std::memcpy(st.data(), pout + 2 * (size_t)B, sizeof(int) * (size_t)B);
  } else {
    CCGP_HIP(hipMemcpyAsync(out_loglik, dll, sizeof(double) * B, hipMemcpyDeviceToHost, h->stream));
    if (out_beta) CCGP_HIP(hipMemcpyAsync(out_beta, dbeta, sizeof(double) * B, hipMemcpyDeviceToHost, h->stream));
    CCGP_HIP(hipMemcpyAsync(st.data(), dst, sizeof(int) * (size_t)B, hipMemcpyDeviceToHost, h->stream));
    CCGP_HIP(hipStreamSynchronize(h->stream));
  }
  if (status) std::memcpy(status, st.data(), sizeof(int) * (size_t)B);
  return count_bad(st.data(), B);
} CCGP_GUARD_END(h)

int ccgp_loglik_grad_batch(ccgp_handle* h, const double* X, int n, int d, const double* y, int K,
                           const double* params, int B, double sigma2, double* out_loglik,
                           double* out_beta, double* out_grad, int* status) try {
  if (!h) return CCGP_EINVAL;
  if (bad_shape(n, d, K) || B < 1 || !X || !y || !params || !out_grad)
    return fail(h, CCGP_EINVAL, "ccgp_loglik_grad_batch: bad argument");
  if (h->fam.id != 0)
    return fail(h, CCGP_EUNSUPPORTED, "ccgp_loglik_grad_batch: analytic gradient is implemented for the Gaussian family only");
  CCGP_HIP(hipSetDevice(h->device));
  const int P = K + K * d;
  if (n > kSmallMaxN || small_lds_bytes(n, d, 1) > (size_t)kLdsBytes - 64) {
    // blocked path: identity rows ride along as extra tile rows, then the tiles of R^-1 are formed (rinv_tile_kernel), turned
    // into M and contracted with the kernel derivatives (grad_contract_kernel; blocked.hip)
    if (!blocked_grad_supported(d, K))
      return fail(h, CCGP_EUNSUPPORTED, "ccgp_loglik_grad_batch: d + K too large for the contraction kernel's LDS");
    const int npad = round_up(n, kTile), nt = npad / kTile, ne = nt;
    const size_t ntiles = blocked_grad_partials(npad);
    const size_t per_extra = sizeof(double) * (ntiles * P + npad);
    size_t per = blocked_ws_bytes(npad, 1, ne) + per_extra;
    size_t glimit = h->ws_limit, gfree = 0, gtotal = 0;
    if (hipMemGetInfo(&gfree, &gtotal) == hipSuccess) {
      const size_t margin = size_t(1) << 30;
      const size_t avail = gfree + h->ws_bytes > margin ? gfree + h->ws_bytes - margin : 0;
      if (avail < glimit) glimit = avail;
    }
    int nbc = (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>((size_t)B, 65535), glimit / per));
    size_t need_st = Carver::al(sizeof(double) * (size_t)n * d) + Carver::al(sizeof(double) * n) +
                     2 * Carver::al(sizeof(double) * (size_t)B * P) + 2 * Carver::al(sizeof(double) * B) +
                     Carver::al(sizeof(int) * (size_t)B);
    int rc = ensure_stage(h, need_st);
    if (rc) return rc;
    rc = ensure_ws(h, blocked_ws_bytes(npad, nbc, ne) + (size_t)nbc * per_extra + 512);
    while (rc == CCGP_ENOMEM && nbc > 1) {   // as in loglik_dev: halve the chunk until the workspace fits
      nbc = (nbc + 1) / 2;
      rc = ensure_ws(h, blocked_ws_bytes(npad, nbc, ne) + (size_t)nbc * per_extra + 512);
    }
    if (rc) return rc;
    Carver c(h->stage);
    double* dX = c.take<double>((size_t)n * d);
    double* dy = c.take<double>(n);
    double* dp = c.take<double>((size_t)B * P);
    double* dg = c.take<double>((size_t)B * P);
    double* dll = c.take<double>(B);
    double* dbeta = c.take<double>(B);
    int* dst = c.take<int>(B);
    CCGP_HIP(hipMemcpyAsync(dX, X, sizeof(double) * (size_t)n * d, hipMemcpyHostToDevice, h->stream));
    CCGP_HIP(hipMemcpyAsync(dy, y, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
    CCGP_HIP(hipMemcpyAsync(dp, params, sizeof(double) * (size_t)B * P, hipMemcpyHostToDevice, h->stream));
    CCGP_HIP(hipMemsetAsync(dst, 0, sizeof(int) * (size_t)B, h->stream));
    DrawView dv{dp, B, K, d};
    dv.fam = h->fam;
    if (int frc = check_family(h, dv.fam, d, K)) return frc;
    Carver tail(static_cast<char*>(h->ws) + Carver::al(blocked_ws_bytes(npad, nbc, ne)));
    BlockedJob job{};
    job.kind = kJobGrad; job.grad = dg; job.Btot = B;
    job.gpart = tail.take<double>((size_t)nbc * ntiles * P);
    job.alpha = tail.take<double>((size_t)nbc * npad);
    for (int b0 = 0; b0 < B; b0 += nbc) {
      const int nb = std::min(nbc, B - b0);
      BlockedWs w = blocked_carve(h->ws, npad, nb, ne);
      blocked_loglik(h, dX, n, d, dy, dv, b0, nb, npad, sigma2, CCGP_MEAN_PROFILE_BETA, 0.0, w, dll, dbeta,
                     dst, &job);
    }
    CCGP_LAUNCH_CHECK();
    std::vector<int> st(B);
    if (out_loglik) CCGP_HIP(hipMemcpyAsync(out_loglik, dll, sizeof(double) * B, hipMemcpyDeviceToHost, h->stream));
    if (out_beta) CCGP_HIP(hipMemcpyAsync(out_beta, dbeta, sizeof(double) * B, hipMemcpyDeviceToHost, h->stream));
    CCGP_HIP(hipMemcpyAsync(out_grad, dg, sizeof(double) * (size_t)B * P, hipMemcpyDeviceToHost, h->stream));
    CCGP_HIP(hipMemcpyAsync(st.data(), dst, sizeof(int) * (size_t)B, hipMemcpyDeviceToHost, h->stream));
    CCGP_HIP(hipStreamSynchronize(h->stream));
    if (status) std::memcpy(status, st.data(), sizeof(int) * (size_t)B);
    return count_bad(st.data(), B);
  }
  const int nch = small_grad_chunks(n, d);
  size_t need = Carver::al(sizeof(double) * (size_t)n * d) + Carver::al(sizeof(double) * n) +
                2 * Carver::al(sizeof(double) * (size_t)B * P) + 2 * Carver::al(sizeof(double) * B) +
                Carver::al(sizeof(int) * (size_t)B) + Carver::al(sizeof(double) * (size_t)B * nch * P);
  int rc = ensure_stage(h, need);
  if (rc) return rc;
  Carver c(h->stage);
  double* dX = c.take<double>((size_t)n * d);
  double* dy = c.take<double>(n);
  double* dp = c.take<double>((size_t)B * P);
  double* dg = c.take<double>((size_t)B * P);
  double* dll = c.take<double>(B);
  double* dbeta = c.take<double>(B);
  int* dst = c.take<int>(B);
  double* dgp = c.take<double>((size_t)B * nch * P);
  if (int prc = push(h, {piece(dX, X, (size_t)n * d), piece(dy, y, n), piece(dp, params, (size_t)B * P)})) return prc;
  DrawView dv{dp, B, K, d};
  dv.fam = h->fam;
  if (int frc = check_family(h, dv.fam, d, K)) return frc;
  {
    ScopedTimer t(h, CCGP_T_FUSED);
    if (small_reg_inverse_supported(n, d, K))
      launch_small_reg_grad(h->stream, dX, n, d, dy, dv, B, sigma2, dll, dbeta, dg, dst);
    else
      launch_small_grad(h->stream, dX, n, d, dy, dv, B, sigma2, dll, dbeta, dg, dst, dgp);
  }
  CCGP_LAUNCH_CHECK();
  std::vector<int> st(B);
  if (int prc = pull(h, {piece(dg, out_grad, (size_t)B * P), piece(dll, out_loglik, B), piece(dbeta, out_beta, B),
                         piece(dst, st.data(), B)}))
    return prc;
  if (status) std::memcpy(status, st.data(), sizeof(int) * (size_t)B);
  return count_bad(st.data(), B);
} CCGP_GUARD_END(h)

// ---- a8: logpost ------------------------------------------------------------------------------
// One transformed parameter vector (psi1, psi2, phi[, zeta]) = (log theta1, log theta2, logit p[, log lambda]) -> the C-ABI
// parameter row (w_1, w_2, theta_1k.., theta_2k..), the log-Jacobian and the script's log-prior (HX:446-463, GV:450, ISO:453,
// ANI:459-462).  theta_t / row: element j of vector b at [b + j * ld]; shared by ccgp_logpost and ccgp_logpost_batch so that a
// value does not depend on which of the two produced it.
static void logpost_terms(int prior_id, int d, const double* theta_t, int ldt, const double* prior_pars, double* row,
                          int ldr, double* log_jacob, double* log_prior) {
  const double psi1 = theta_t[0], psi2 = theta_t[ldt], phi = theta_t[2 * (size_t)ldt];
  const double theta1 = std::exp(psi1), theta2 = std::exp(psi2);
  const double p = 1.0 / (1.0 + std::exp(-phi));
  row[0] = p;
  row[ldr] = 1.0 - p;
  double lj = -phi - 2.0 * std::log(1.0 + std::exp(-phi)) + psi1 + psi2;
  double lp = 0.0;
  if (prior_id == CCGP_PRIOR_ANI) {
    const double zeta = theta_t[3 * (size_t)ldt], lambda = std::exp(zeta);
    row[2 * (size_t)ldr] = theta1; row[3 * (size_t)ldr] = theta2;
    row[4 * (size_t)ldr] = (1.0 + lambda) * theta1; row[5 * (size_t)ldr] = (1.0 + lambda) * theta2;
    lj += zeta;
    lp = -psi1 - psi1 * psi1 / 2.0 - psi2 - psi2 * psi2 / 2.0 - 4.0 * zeta - 4.0 / lambda;
  } else {
    for (int k = 0; k < d; ++k) { row[(size_t)(2 + k) * ldr] = theta1; row[(size_t)(2 + d + k) * ldr] = theta2; }
    if (prior_id == CCGP_PRIOR_INVGAMMA)
      lp = -(prior_pars[0] + 1.0) * psi1 - prior_pars[1] / theta1 -
           (prior_pars[2] + 1.0) * psi2 - prior_pars[3] / theta2;
    else if (prior_id == CCGP_PRIOR_GV)
      lp = -4.0 * psi1 - 1.0 / theta1 - 6.0 * psi2 - 75.0 / theta2;
    else
      lp = -4.0 * psi1 - 2.0 / theta1 - 6.0 * psi2 - 16.0 / theta2;
  }
  *log_jacob = lj;
  *log_prior = lp;
}

int ccgp_logpost_batch(ccgp_handle* h, const double* X, int n, int d, const double* y, double sigma2, int prior_id,
                       const double* theta_t, int B, const double* prior_pars, double* out_val, double* out_beta,
                       double* out_loglik, int* status) try {
  if (!h) return CCGP_EINVAL;
  if (n < 1 || d < 1 || d > kMaxD || B < 1 || !X || !y || !theta_t || !out_val)
    return fail(h, CCGP_EINVAL, "ccgp_logpost_batch: bad argument");
  if (prior_id < CCGP_PRIOR_INVGAMMA || prior_id > CCGP_PRIOR_ANI)
    return fail(h, CCGP_EINVAL, "ccgp_logpost_batch: unknown prior_id");
  if (prior_id == CCGP_PRIOR_INVGAMMA && !prior_pars)
    return fail(h, CCGP_EINVAL, "ccgp_logpost_batch: prior_pars required for CCGP_PRIOR_INVGAMMA");
  if (prior_id == CCGP_PRIOR_ANI && d != 2)
    return fail(h, CCGP_EINVAL, "ccgp_logpost_batch: the anisotropic script (ANI) is 2-D");
  const int K = 2, P = K + K * d;
  std::vector<double> rows((size_t)B * P), ljac(B), lpri(B), ll(B), beta(B);
  std::vector<int> st(B);
  for (int b = 0; b < B; ++b) {
    double lj = 0.0, lp = 0.0;
    logpost_terms(prior_id, d, theta_t + b, B, prior_pars, rows.data() + b, B, &lj, &lp);
    ljac[b] = lj;
    lpri[b] = lp;
  }
  const int rc = ccgp_loglik_batch(h, X, n, d, y, K, rows.data(), B, sigma2, CCGP_MEAN_PROFILE_BETA, 0.0, ll.data(), beta.data(),
                                   st.data());
  if (rc < 0) return rc;
  for (int b = 0; b < B; ++b) {
    out_val[b] = ll[b] + ljac[b] + lpri[b];   // ccgp_logpost's order of additions; NaN where the factorisation failed: the reference's NA
    if (out_beta) out_beta[b] = beta[b];
    if (out_loglik) out_loglik[b] = ll[b];
    if (status) status[b] = st[b];
  }
  return rc;
} CCGP_GUARD_END(h)

int ccgp_logpost(ccgp_handle* h, const double* X, int n, int d, const double* y, double sigma2,
                 int prior_id, const double* theta_t, const double* prior_pars, double* out_val,
                 double* out_beta, double* out_loglik, double* out_Rinv, int* status) try {
  if (!h) return CCGP_EINVAL;
  if (n < 1 || d < 1 || d > kMaxD || !X || !y || !theta_t || !out_val)
    return fail(h, CCGP_EINVAL, "ccgp_logpost: bad argument");
  if (prior_id < CCGP_PRIOR_INVGAMMA || prior_id > CCGP_PRIOR_ANI)
    return fail(h, CCGP_EINVAL, "ccgp_logpost: unknown prior_id");
  if (prior_id == CCGP_PRIOR_INVGAMMA && !prior_pars)
    return fail(h, CCGP_EINVAL, "ccgp_logpost: prior_pars required for CCGP_PRIOR_INVGAMMA");
  if (prior_id == CCGP_PRIOR_ANI && d != 2)
    return fail(h, CCGP_EINVAL, "ccgp_logpost: the anisotropic script (ANI) is 2-D");
  const int K = 2, P = K + K * d;
  std::vector<double> row(P);
  double log_jacob = 0.0, log_prior = 0.0;
  logpost_terms(prior_id, d, theta_t, 1, prior_pars, row.data(), 1, &log_jacob, &log_prior);
  // ONE factorisation per call (the reference runs two: solve(R) at HX:454 and the Cholesky inside dmnorm at
  // HX:460): without R.Inv the batched evaluator; with it, a sweep that carries y', 1' AND the identity rows,
  // so the likelihood, beta and R^-1 come out of the same elimination
  double ll = 0.0, beta = 0.0;
  int st = 0;
  CCGP_HIP(hipSetDevice(h->device));
  const bool gauss = h->fam.id == 0;
  const bool reg_val = gauss && small_reg_supported(n, d, K);
  const bool reg_inv = gauss && small_reg_inverse_supported(n, d, K);
  const size_t in_d = (size_t)n * d + n + P;                       // X | y | row
  const size_t out_d = 3 + (out_Rinv ? (size_t)n * n : 0);         // ll, beta, status (as one double slot) | R^-1
  if ((out_Rinv ? reg_inv : reg_val) && ensure_pin(h, sizeof(double) * (in_d + out_d)) == CCGP_OK) {
    // The sequential caller's path (Metro evaluates ONE proposal per logpost call, HX:505-512): latency, not
    // throughput.  Inputs are packed into a pinned host buffer and cross PCIe in ONE copy, the results (log-lik,
    // beta, status[, R^-1]) come back in one: 300 -> ~100 us per call with R.Inv at n = 64, 113 -> ~60 us without (round 2).
    int rc2 = ensure_stage(h, Carver::al(sizeof(double) * in_d) + Carver::al(sizeof(double) * out_d) + 256);
    if (rc2) return rc2;
    double* pin = static_cast<double*>(h->pin);
    std::memcpy(pin, X, sizeof(double) * (size_t)n * d);
    std::memcpy(pin + (size_t)n * d, y, sizeof(double) * n);
    std::memcpy(pin + (size_t)n * d + n, row.data(), sizeof(double) * P);
    Carver c(h->stage);
    // With R.Inv the kernel stores its results STRAIGHT into the pinned host buffer (device-visible like any hipHostMalloc
    // memory): no device-to-host copy of the n x n inverse behind the kernel -- 84 -> 72 us per call at n = 64, 125 -> 115 at
    // n = 90.  (Value only: three doubles, no difference; reading the INPUTS through PCIe from the kernel is slower.)
    const bool zo = out_Rinv != nullptr;
    double* pout = pin + in_d;
    double* din = c.take<double>(in_d);
    double* dout = zo ? pout : c.take<double>(out_d);
    double* dX = din;
    double* dy = din + (size_t)n * d;
    double* dp = dy + n;
    int* dst = reinterpret_cast<int*>(dout + 2);
    CCGP_HIP(hipMemcpyAsync(din, pin, sizeof(double) * in_d, hipMemcpyHostToDevice, h->stream));
    DrawView dv{dp, 1, K, d};
    dv.fam = h->fam;
    {
      ScopedTimer t(h, CCGP_T_FUSED);
      if (out_Rinv)
        launch_small_reg_inverse(h->stream, dX, n, d, dy, dv, 0, sigma2, dout + 3, dout, dout + 1, dst);
      else
        launch_small_reg_loglik(h->stream, dX, n, d, dy, dv, 1, sigma2, CCGP_MEAN_PROFILE_BETA, 0.0, dout, dout + 1, dst);
    }
    CCGP_LAUNCH_CHECK();
    if (!zo) CCGP_HIP(hipMemcpyAsync(pout, dout, sizeof(double) * out_d, hipMemcpyDeviceToHost, h->stream));
    CCGP_HIP(hipStreamSynchronize(h->stream));
    ll = pout[0];
    beta = pout[1];
    std::memcpy(&st, pout + 2, sizeof(int));
    if (out_Rinv) std::memcpy(out_Rinv, pout + 3, sizeof(double) * (size_t)n * n);
  } else if (!out_Rinv) {
    int rc = ccgp_loglik_batch(h, X, n, d, y, K, row.data(), 1, sigma2, CCGP_MEAN_PROFILE_BETA, 0.0, &ll, &beta,
                               &st);
    if (rc < 0) return rc;
  } else {
    const bool blocked = h->fam.id != 0 || n > kSmallMaxN || small_lds_bytes(n, d, 1) > (size_t)kLdsBytes - 64;
    const int npad = round_up(n, kTile), nt = npad / kTile;
    size_t need = Carver::al(sizeof(double) * (size_t)n * d) + Carver::al(sizeof(double) * n) +
                  Carver::al(sizeof(double) * P) + Carver::al(sizeof(double) * (size_t)n * n) +
                  3 * Carver::al(sizeof(double) * 2) + 256;
    int rc2 = ensure_stage(h, need);
    if (rc2) return rc2;
    if (blocked) {
      rc2 = ensure_ws(h, blocked_ws_bytes(npad, 1, nt));
      if (rc2) return rc2;
    }
    Carver c(h->stage);
    double* dX = c.take<double>((size_t)n * d);
    double* dy = c.take<double>(n);
    double* dp = c.take<double>(P);
    double* dR = c.take<double>((size_t)n * n);
    double* dll = c.take<double>(1);
    double* dbt = c.take<double>(1);
    int* dst = c.take<int>(1);
    CCGP_HIP(hipMemcpyAsync(dX, X, sizeof(double) * (size_t)n * d, hipMemcpyHostToDevice, h->stream));
    CCGP_HIP(hipMemcpyAsync(dy, y, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
    CCGP_HIP(hipMemcpyAsync(dp, row.data(), sizeof(double) * P, hipMemcpyHostToDevice, h->stream));
    CCGP_HIP(hipMemsetAsync(dst, 0, sizeof(int), h->stream));
    DrawView dv{dp, 1, K, d};
    dv.fam = h->fam;
    if (int frc = check_family(h, dv.fam, d, K)) return frc;
    if (blocked) {
      // identity as extra tile rows of the blocked sweep, then R^-1 = Z Z' tile by tile
      BlockedJob job{};
      job.kind = kJobInverse; job.Rinv = dR;
      BlockedWs w = blocked_carve(h->ws, npad, 1, nt);
      blocked_loglik(h, dX, n, d, dy, dv, 0, 1, npad, sigma2, CCGP_MEAN_PROFILE_BETA, 0.0, w, dll, dbt, dst, &job);
    } else {
      ScopedTimer t(h, CCGP_T_FUSED);
      launch_small_inverse(h->stream, dX, n, d, dy, dv, 0, sigma2, dR, dll, dbt, dst);
    }
    CCGP_LAUNCH_CHECK();
    CCGP_HIP(hipMemcpyAsync(out_Rinv, dR, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToHost, h->stream));
    CCGP_HIP(hipMemcpyAsync(&ll, dll, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    CCGP_HIP(hipMemcpyAsync(&beta, dbt, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    CCGP_HIP(hipMemcpyAsync(&st, dst, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    CCGP_HIP(hipStreamSynchronize(h->stream));
  }
  *out_val = ll + log_jacob + log_prior;
  if (out_beta) *out_beta = beta;
  if (out_loglik) *out_loglik = ll;
  if (status) *status = st;
  return st != 0 ? 1 : 0;
} CCGP_GUARD_END(h)

// ---- 8(f)-4: entropy criteria over candidate designs ------------------------------------------------
int ccgp_mixed_logdet_designs(ccgp_handle* h, const double* Xs, int n, int d, int B, int K,
                              const double* params, double* out_logdet, int* status) try {
  if (!h) return CCGP_EINVAL;
  if (bad_shape(n, d, K) || B < 1 || !Xs || !params || !out_logdet)
    return fail(h, CCGP_EINVAL, "ccgp_mixed_logdet_designs: bad argument");
  if (h->fam.id != 0)
    return fail(h, CCGP_EUNSUPPORTED, "ccgp_mixed_logdet_designs: Gaussian family only (BSQ:856-877)");
  CCGP_HIP(hipSetDevice(h->device));
  const int P = K + K * d;
  if (!small_reg_supported(n, d, K, true)) {
    // more than 128 points (or too wide for the register-resident evaluator): the blocked sweep, one design at a time --
    // its chunk shares ONE design among its matrices, and here every matrix has its own.  The reference's candidate
    // sets are small (BSQ:856-877: a few dozen points); this branch exists so that the entry point has no size limit.
    const int npad = round_up(n, kTile);
    size_t need = Carver::al(sizeof(double) * (size_t)B * n * d) + Carver::al(sizeof(double) * n) + Carver::al(sizeof(double) * P) +
                  3 * Carver::al(sizeof(double) * B) + Carver::al(sizeof(int) * (size_t)B);
    int rc = ensure_stage(h, need);
    if (rc) return rc;
    rc = ensure_ws(h, blocked_ws_bytes(npad, 1, 0) + 512);
    if (rc) return rc;
    Carver c(h->stage);
    double* dXs = c.take<double>((size_t)B * n * d);
    double* dy = c.take<double>(n);
    double* dp = c.take<double>(P);
    double* dld = c.take<double>(B);
    double* dll = c.take<double>(B);
    double* dbeta = c.take<double>(B);
    int* dst = c.take<int>(B);
    if (int prc = push(h, {piece(dXs, Xs, (size_t)B * n * d), piece(dp, params, P)})) return prc;
    CCGP_HIP(hipMemsetAsync(dy, 0, sizeof(double) * n, h->stream));
    CCGP_HIP(hipMemsetAsync(dst, 0, sizeof(int) * (size_t)B, h->stream));
    DrawView dv{dp, 1, K, d};
    dv.fam = h->fam;
    if (int frc = check_family(h, dv.fam, d, K)) return frc;
    BlockedWs w = blocked_carve(h->ws, npad, 1, 0);
    for (int i = 0; i < B; ++i) {
      BlockedJob job{};
      job.kind = kJobLogdet;
      job.logdet = dld + i;
      blocked_loglik(h, dXs + (size_t)i * n * d, n, d, dy, dv, 0, 1, npad, 1.0, CCGP_MEAN_PROFILE_BETA, 0.0, w, dll + i,
                     dbeta + i, dst + i, &job);
    }
    CCGP_LAUNCH_CHECK();
    std::vector<int> st(B);
    if (int prc = pull(h, {piece(dld, out_logdet, B), piece(dst, st.data(), B)})) return prc;
    if (status) std::memcpy(status, st.data(), sizeof(int) * (size_t)B);
    return count_bad(st.data(), B);
  }
  size_t need = Carver::al(sizeof(double) * (size_t)B * n * d) + Carver::al(sizeof(double) * P) +
                Carver::al(sizeof(double) * B) + Carver::al(sizeof(int) * (size_t)B);
  int rc = ensure_stage(h, need);
  if (rc) return rc;
  Carver c(h->stage);
  double* dXs = c.take<double>((size_t)B * n * d);
  double* dp = c.take<double>(P);
  double* dld = c.take<double>(B);
  int* dst = c.take<int>(B);
  if (int prc = push(h, {piece(dXs, Xs, (size_t)B * n * d), piece(dp, params, P)})) return prc;
  DrawView dv{dp, 1, K, d};
  dv.fam = h->fam;
  if (int frc = check_family(h, dv.fam, d, K)) return frc;
  {
    ScopedTimer t(h, CCGP_T_FUSED);
    launch_small_reg_logdet_designs(h->stream, dXs, n, d, dv, B, dld, dst);
  }
  CCGP_LAUNCH_CHECK();
  std::vector<int> st(B);
  if (int prc = pull(h, {piece(dld, out_logdet, B), piece(dst, st.data(), B)})) return prc;
  if (status) std::memcpy(status, st.data(), sizeof(int) * (size_t)B);
  return count_bad(st.data(), B);
} CCGP_GUARD_END(h)

// ---- a9: hyperprior grid ------------------------------------------------------------------------
int ccgp_halton_base2(int N, double* out) {
  if (N < 0 || !out) return CCGP_EINVAL;
  halton_base2(N, out);
  return CCGP_OK;
}

int ccgp_qigamma(const double* p, int N, double alpha, double beta, double* out) {
  if (N < 0 || !p || !out || !(alpha > 0.0) || !(beta > 0.0)) return CCGP_EINVAL;
  for (int i = 0; i < N; ++i) out[i] = qigamma_host(p[i], alpha, beta);
  return CCGP_OK;
}

int ccgp_grid_marginal(ccgp_handle* h, const double* X, int n, int d, const double* y,
                       double sigma2, const double* hyper, int G, int N, double tau, int take_log,
                       double aniso_lambda, double* out, int* out_argmax, double* out_logs) try {
  if (!h) return CCGP_EINVAL;
  if (n < 1 || d < 1 || d > kMaxD || G < 1 || N < 1 || !X || !y || !hyper || !out)
    return fail(h, CCGP_EINVAL, "ccgp_grid_marginal: bad argument");
  const bool aniso = aniso_lambda >= 0.0;
  if (aniso && d != 2) return fail(h, CCGP_EINVAL, "ccgp_grid_marginal: anisotropic kernel is 2-D");
  CCGP_HIP(hipSetDevice(h->device));
  const int K = 2, P = K + K * d;
  const size_t B = (size_t)G * N;
  if (B > (size_t)1 << 30) return fail(h, CCGP_EINVAL, "ccgp_grid_marginal: G*N too large");
  // What crosses PCIe: X, y, the G x 4 hyperparameter matrix and the list of DISTINCT inverse-gamma shapes
  // (a few KB) in; G doubles and one failure count out.  The G x N table of draws is built in HBM:
  // unit-rate gamma quantiles per distinct shape (grid_qtab_kernel; the same Halton node drives p, theta1 and
  // theta2, HX:554-556), then the parameter rows (grid_expand_kernel).
  std::vector<double> shapes;
  std::vector<int> shape_idx(2 * (size_t)G);
  {
    std::map<double, int> seen;
    for (int g = 0; g < G; ++g) {
      const double a1 = hyper[g], b1 = hyper[g + (size_t)G], a2 = hyper[g + (size_t)2 * G],
                   b2 = hyper[g + (size_t)3 * G];
      if (!(a1 > 0.0) || !(b1 > 0.0) || !(a2 > 0.0) || !(b2 > 0.0))
        return fail(h, CCGP_EINVAL, "ccgp_grid_marginal: hyperparameters must be positive");
      for (int w = 0; w < 2; ++w) {
        const double al = w ? a2 : a1;
        auto it = seen.find(al);
        if (it == seen.end()) {
          it = seen.emplace(al, (int)shapes.size()).first;
          shapes.push_back(al);
        }
        shape_idx[(size_t)w * G + g] = it->second;
      }
    }
  }
  const size_t ns = shapes.size();
  size_t need = Carver::al(sizeof(double) * (size_t)n * d) + Carver::al(sizeof(double) * n) +
                Carver::al(sizeof(double) * 4 * G) + Carver::al(sizeof(double) * ns) +
                Carver::al(sizeof(int) * 2 * G) + Carver::al(sizeof(double) * ns * N) +
                Carver::al(sizeof(double) * B * P) + 2 * Carver::al(sizeof(double) * B) +
                Carver::al(sizeof(int) * B) + Carver::al(sizeof(double) * G) + Carver::al(sizeof(int));
  int rc = ensure_stage(h, need);
  if (rc) return rc;
  Carver c(h->stage);
  double* dX = c.take<double>((size_t)n * d);
  double* dy = c.take<double>(n);
  double* dhy = c.take<double>((size_t)4 * G);
  double* dsh = c.take<double>(ns);
  int* dsi = c.take<int>((size_t)2 * G);
  double* dq = c.take<double>(ns * N);
  double* dp = c.take<double>(B * P);
  double* dll = c.take<double>(B);
  double* dbeta = c.take<double>(B);
  int* dst = c.take<int>(B);
  double* dout = c.take<double>(G);
  int* dbad = c.take<int>(1);
  CCGP_HIP(hipMemcpyAsync(dX, X, sizeof(double) * (size_t)n * d, hipMemcpyHostToDevice, h->stream));
  CCGP_HIP(hipMemcpyAsync(dy, y, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
  CCGP_HIP(hipMemcpyAsync(dhy, hyper, sizeof(double) * 4 * (size_t)G, hipMemcpyHostToDevice, h->stream));
  CCGP_HIP(hipMemcpyAsync(dsh, shapes.data(), sizeof(double) * ns, hipMemcpyHostToDevice, h->stream));
  CCGP_HIP(hipMemcpyAsync(dsi, shape_idx.data(), sizeof(int) * 2 * (size_t)G, hipMemcpyHostToDevice, h->stream));
  CCGP_HIP(hipMemsetAsync(dbad, 0, sizeof(int), h->stream));
  {
    ScopedTimer t(h, CCGP_T_COV);
    // ns x N quantiles (a dozen shapes x 1000 nodes: ~12 000 threads, each a long serial Halley iteration) cannot fill
    // the chip; what matters is each wave's own speed.  One wave per workgroup spreads them over ~190 CUs where a lone
    // wave issues every 7.5 cycles, against one instruction per ~19 cycles when four waves share a SIMD
    // (profiles/r04/valu_f64_rates.txt): 1.96 -> 1.17 ms of the Heat-Exchanger grid's 11 ms end to end (and 0.08 once the quantile iteration stops at rounding level: special_math.h)
    hipLaunchKernelGGL(grid_qtab_kernel, dim3((unsigned)((ns * N + 63) / 64)), dim3(64), 0, h->stream, dsh, (int)ns, N, dq);
    hipLaunchKernelGGL(grid_expand_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, h->stream, dhy, dsi, dq, G, N,
                       d, aniso ? 1 : 0, aniso_lambda, dp);
  }
  rc = loglik_dev(h, dX, n, d, dy, K, dp, (int)B, sigma2, CCGP_MEAN_ZERO_PLUS_TAU2, tau * tau, dll,
                  dbeta, dst);
  if (rc) return rc;
  hipLaunchKernelGGL(row_logmeanexp_kernel, dim3(G), dim3(256), 0, h->stream, dll, N, take_log, dout);
  hipLaunchKernelGGL(count_bad_kernel, dim3((unsigned)((B + 1023) / 1024)), dim3(256), 0, h->stream, dst, (int)B, dbad);
  CCGP_LAUNCH_CHECK();
  int bad = 0;
  CCGP_HIP(hipMemcpyAsync(out, dout, sizeof(double) * G, hipMemcpyDeviceToHost, h->stream));
  CCGP_HIP(hipMemcpyAsync(&bad, dbad, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  if (out_logs) CCGP_HIP(hipMemcpyAsync(out_logs, dll, sizeof(double) * B, hipMemcpyDeviceToHost, h->stream));
  CCGP_HIP(hipStreamSynchronize(h->stream));
  if (out_argmax) {
    // which.max: first maximum, NaN skipped
    int best = -1;
    for (int g = 0; g < G; ++g)
      if (out[g] == out[g] && (best < 0 || out[g] > out[best])) best = g;
    *out_argmax = best;
  }
  return bad;
} CCGP_GUARD_END(h)

// ---- a10/a11: prediction -------------------------------------------------------------------------
int ccgp_predict_batch_dev(ccgp_handle* h, const double* dX, int n, int d, const double* dy, int K,
                           const double* dparams, int S, const double* dXtest, int m,
                           double sigma2, double* d_mean, double* d_var, double* d_beta,
                           int* d_status) try {
  if (!h) return CCGP_EINVAL;
  if (bad_shape(n, d, K) || S < 1 || m < 1 || !dX || !dy || !dparams || !dXtest || !d_mean || !d_var)
    return fail(h, CCGP_EINVAL, "ccgp_predict_batch: bad argument");
  CCGP_HIP(hipSetDevice(h->device));
  DrawView dv{dparams, S, K, d};
  dv.fam = h->fam;
  if (int frc = check_family(h, dv.fam, d, K)) return frc;
  if (dv.fam.id != 0 || n > kSmallMaxN || small_lds_bytes(n, d, 1) > (size_t)kLdsBytes - 64) {
    // blocked path: the m cross-correlation rows ride along as extra tile rows of the sweep
    const int npad = round_up(n, kTile), ne = (m + kTile - 1) / kTile;
    int nbc = blocked_chunk(h, npad, S, ne);
    size_t extra = Carver::al(sizeof(double) * (size_t)S) * 2 + Carver::al(sizeof(int) * (size_t)S);
    int rc = ensure_ws(h, blocked_ws_bytes(npad, nbc, ne) + extra);
    while (rc == CCGP_ENOMEM && nbc > 1) {   // another handle / process took the memory in between: smaller chunks
      nbc = (nbc + 1) / 2;
      rc = ensure_ws(h, blocked_ws_bytes(npad, nbc, ne) + extra);
    }
    if (rc) return rc;
    // scratch for outputs the caller did not ask for lives behind the matrices
    Carver tail(static_cast<char*>(h->ws) + blocked_ws_bytes(npad, nbc, ne));
    double* ll = tail.take<double>(S);
    double* bt = d_beta ? d_beta : tail.take<double>(S);
    int* st = d_status ? d_status : tail.take<int>(S);
    CCGP_HIP(hipMemsetAsync(st, 0, sizeof(int) * (size_t)S, h->stream));
    BlockedJob pr{};
    pr.kind = kJobPredict; pr.Xtest = dXtest; pr.m = m; pr.S = S; pr.mean = d_mean; pr.var = d_var;
    for (int b0 = 0; b0 < S; b0 += nbc) {
      const int nb = std::min(nbc, S - b0);
      BlockedWs w = blocked_carve(h->ws, npad, nb, ne);
      blocked_loglik(h, dX, n, d, dy, dv, b0, nb, npad, sigma2, CCGP_MEAN_PROFILE_BETA, 0.0, w, ll, bt, st,
                     &pr);
    }
    CCGP_LAUNCH_CHECK();
    return CCGP_OK;
  }
  {
    ScopedTimer t(h, CCGP_T_FUSED);
    if (small_reg_supported(n, d, K, false, true)) {
      // kept-factor scheme where it applies (n <= 104, K <= 3) and its scratch fits the workspace: the factor block and the
      // correlation vectors of as many draws at a time as the limit allows (at least 64)
      void* scratch = nullptr;
      size_t sbytes = 0;
      if (h->opt_predict_factor && small_reg_sites_supported(n, d, K)) {
        const size_t per = small_reg_sites_scratch(n, d, K, m);
        const size_t want = per * (size_t)S, cap = std::max<size_t>(h->ws_limit / 2, per * 64);
        if (ensure_ws(h, std::min(want, cap / per * per)) == CCGP_OK) { scratch = h->ws; sbytes = h->ws_bytes; }
        if (scratch && !h->aux_stream) {
          if (hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking) != hipSuccess) h->aux_stream = nullptr;
          if (h->aux_stream && (hipEventCreateWithFlags(&h->aux_fork, hipEventDisableTiming) != hipSuccess ||
                                hipEventCreateWithFlags(&h->aux_join, hipEventDisableTiming) != hipSuccess)) {
            (void)hipStreamDestroy(h->aux_stream);
            h->aux_stream = nullptr;
          }
        }
      }
      launch_small_reg_predict(h->stream, dX, n, d, dy, dv, S, dXtest, m, sigma2, d_mean, d_var, d_beta,
                               d_status, scratch, sbytes, h->aux_stream, h->aux_fork, h->aux_join);
    } else
      launch_small_predict(h->stream, dX, n, d, dy, dv, S, dXtest, m, sigma2, d_mean, d_var, d_beta,
                           d_status);
  }
  CCGP_LAUNCH_CHECK();
  return CCGP_OK;
} CCGP_GUARD_END(h)

int ccgp_predict_batch(ccgp_handle* h, const double* X, int n, int d, const double* y, int K,
                       const double* params, int S, const double* Xtest, int m, double sigma2,
                       double* out_mean, double* out_var, double* out_beta, int* status) try {
  if (!h) return CCGP_EINVAL;
  if (bad_shape(n, d, K) || S < 1 || m < 1 || !X || !y || !params || !Xtest || !out_mean || !out_var)
    return fail(h, CCGP_EINVAL, "ccgp_predict_batch: bad argument");
  CCGP_HIP(hipSetDevice(h->device));
  const int P = K + K * d;
  size_t need = Carver::al(sizeof(double) * (size_t)n * d) + Carver::al(sizeof(double) * n) +
                Carver::al(sizeof(double) * (size_t)S * P) + Carver::al(sizeof(double) * (size_t)m * d) +
                2 * Carver::al(sizeof(double) * (size_t)S * m) + Carver::al(sizeof(double) * S) +
                Carver::al(sizeof(int) * (size_t)S);
  int rc = ensure_stage(h, need);
  if (rc) return rc;
  Carver c(h->stage);
  double* dX = c.take<double>((size_t)n * d);
  double* dy = c.take<double>(n);
  double* dp = c.take<double>((size_t)S * P);
  double* dXt = c.take<double>((size_t)m * d);
  double* dmean = c.take<double>((size_t)S * m);
  double* dvar = c.take<double>((size_t)S * m);
  double* dbeta = c.take<double>(S);
  int* dst = c.take<int>(S);
  if (int prc = push(h, {piece(dX, X, (size_t)n * d), piece(dy, y, n), piece(dp, params, (size_t)S * P),
                         piece(dXt, Xtest, (size_t)m * d)}))
    return prc;
  rc = ccgp_predict_batch_dev(h, dX, n, d, dy, K, dp, S, dXt, m, sigma2, dmean, dvar, dbeta, dst);
  if (rc) return rc;
  std::vector<int> st(S);
  if (int prc = pull(h, {piece(dmean, out_mean, (size_t)S * m), piece(dvar, out_var, (size_t)S * m),
                         piece(dbeta, out_beta, S), piece(dst, st.data(), S)}))
    return prc;
  if (status) std::memcpy(status, st.data(), sizeof(int) * (size_t)S);
  return count_bad(st.data(), S);
} CCGP_GUARD_END(h)

// ---- 8(f)-2: device-resident factor set --------------------------------------------------------------
struct ccgp_factorset {
  int device = 0;
  int n = 0, d = 0, K = 0, S = 0, npad = 0;
  double sigma2 = 0.0;
  ccgp::KernelFamily fam;
  bool fused = false;      // n <= 128: nothing but the draws is kept (see ccgp_factor_batch)
  void* mem = nullptr;
  size_t bytes = 0;
  double* X = nullptr;     // device copies
  double* y = nullptr;
  double* params = nullptr;
  double* ll = nullptr;
  double* beta = nullptr;
  int* status = nullptr;
  ccgp::BlockedWs w{};
};

int ccgp_factor_batch(ccgp_handle* h, const double* X, int n, int d, const double* y, int K,
                      const double* params, int S, double sigma2, ccgp_factorset** out,
                      double* out_loglik, double* out_beta, int* status) try {
  if (!h) return CCGP_EINVAL;
  if (bad_shape(n, d, K) || S < 1 || !X || !y || !params || !out)
    return fail(h, CCGP_EINVAL, "ccgp_factor_batch: bad argument");
  *out = nullptr;
  if (int frc = check_family(h, h->fam, d, K)) return frc;
  CCGP_HIP(hipSetDevice(h->device));
  const int P = K + K * d;
  const bool fused = h->fam.id == 0 && n <= kSmallMaxN && small_lds_bytes(n, d, 1) <= (size_t)kLdsBytes - 64;
  const int npad = round_up(n, kTile);
  if (!fused && S > 65535)   // the draw index is a grid y / z dimension in cov_kernel, rhs_rows_kernel, ...
    return fail(h, CCGP_EINVAL, "ccgp_factor_batch: at most 65535 factors per set on the blocked path (n > 128)");
  size_t head = Carver::al(sizeof(double) * (size_t)n * d) + Carver::al(sizeof(double) * n) +
                Carver::al(sizeof(double) * (size_t)S * P) + 2 * Carver::al(sizeof(double) * S) +
                Carver::al(sizeof(int) * (size_t)S);
  size_t bytes = head + (fused ? 0 : blocked_ws_bytes(npad, S, 0) + 256);
  ccgp_factorset* fs = new ccgp_factorset();
  if (hipMalloc(&fs->mem, bytes) != hipSuccess) {
    delete fs;
    return fail(h, CCGP_ENOMEM, "ccgp_factor_batch: " + std::to_string(bytes) + " B for " + std::to_string(S) +
                                    " factors do not fit on the device");
  }
  fs->bytes = bytes; fs->device = h->device; fs->n = n; fs->d = d; fs->K = K; fs->S = S; fs->npad = npad;
  fs->sigma2 = sigma2; fs->fam = h->fam; fs->fused = fused;
  Carver c(fs->mem);
  fs->X = c.take<double>((size_t)n * d);
  fs->y = c.take<double>(n);
  fs->params = c.take<double>((size_t)S * P);
  fs->ll = c.take<double>(S);
  fs->beta = c.take<double>(S);
  fs->status = c.take<int>(S);
  auto bail = [&](int code) {
    (void)hipFree(fs->mem);
    delete fs;
    return code;
  };
  if (hipMemcpyAsync(fs->X, X, sizeof(double) * (size_t)n * d, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
      hipMemcpyAsync(fs->y, y, sizeof(double) * n, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
      hipMemcpyAsync(fs->params, params, sizeof(double) * (size_t)S * P, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
      hipMemsetAsync(fs->status, 0, sizeof(int) * (size_t)S, h->stream) != hipSuccess)
    return bail(fail(h, CCGP_EHIP, "ccgp_factor_batch: upload failed"));
  DrawView dv{fs->params, S, K, d};
  dv.fam = h->fam;
  if (fused) {
    // n <= 128: the factor of a draw lives and dies in registers / LDS inside the fused evaluator (10 us);
    // storing it would cost more HBM traffic than regenerating it.  The set keeps the draws; likelihood
    // and beta are evaluated once here, prediction re-runs the fused predictor on the resident inputs.
    int rc = loglik_dev(h, fs->X, n, d, fs->y, K, fs->params, S, sigma2, CCGP_MEAN_PROFILE_BETA, 0.0, fs->ll,
                        fs->beta, fs->status);
    if (rc) return bail(rc);
  } else {
    fs->w = blocked_carve(c.take<char>(blocked_ws_bytes(npad, S, 0)), npad, S, 0);
    blocked_loglik(h, fs->X, n, d, fs->y, dv, 0, S, npad, sigma2, CCGP_MEAN_PROFILE_BETA, 0.0, fs->w, fs->ll,
                   fs->beta, fs->status);
    if (hipGetLastError() != hipSuccess) return bail(fail(h, CCGP_EHIP, "ccgp_factor_batch: launch failed"));
    {
      const std::string ae = ccgp::attr_error(h->device);
      if (!ae.empty()) return bail(fail(h, CCGP_EHIP, ae));
    }
  }
  std::vector<int> st(S);
  hipError_t e = hipSuccess;
  if (out_loglik) e = hipMemcpyAsync(out_loglik, fs->ll, sizeof(double) * S, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess && out_beta)
    e = hipMemcpyAsync(out_beta, fs->beta, sizeof(double) * S, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(st.data(), fs->status, sizeof(int) * (size_t)S, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) return bail(fail(h, CCGP_EHIP, std::string("ccgp_factor_batch: ") + hipGetErrorString(e)));
  if (status) std::memcpy(status, st.data(), sizeof(int) * (size_t)S);
  *out = fs;
  return count_bad(st.data(), S);
} CCGP_GUARD_END(h)

int ccgp_predict_from_factorset(ccgp_handle* h, const ccgp_factorset* fs, const double* Xtest, int m,
                                double* out_mean, double* out_var) try {
  if (!h) return CCGP_EINVAL;
  if (!fs || !Xtest || m < 1 || !out_mean || !out_var)
    return fail(h, CCGP_EINVAL, "ccgp_predict_from_factorset: bad argument");
  if (fs->device != h->device)
    return fail(h, CCGP_EINVAL, "ccgp_predict_from_factorset: the factor set lives on another device");
  CCGP_HIP(hipSetDevice(h->device));
  const int n = fs->n, d = fs->d, S = fs->S;
  size_t need = Carver::al(sizeof(double) * (size_t)m * d) + 2 * Carver::al(sizeof(double) * (size_t)S * m);
  int rc = ensure_stage(h, need);
  if (rc) return rc;
  Carver c(h->stage);
  double* dXt = c.take<double>((size_t)m * d);
  double* dmean = c.take<double>((size_t)S * m);
  double* dvar = c.take<double>((size_t)S * m);
  CCGP_HIP(hipMemcpyAsync(dXt, Xtest, sizeof(double) * (size_t)m * d, hipMemcpyHostToDevice, h->stream));
  if (fs->fused) {
    const ccgp::KernelFamily keep = h->fam;
    h->fam = fs->fam;
    rc = ccgp_predict_batch_dev(h, fs->X, n, d, fs->y, fs->K, fs->params, S, dXt, m, fs->sigma2, dmean, dvar,
                                nullptr, nullptr);
    h->fam = keep;
    if (rc) return rc;
  } else {
    const int ne = (m + kTile - 1) / kTile, lde = ne * kTile;
    const size_t e_stride = (size_t)lde * fs->npad;
    // the m cross-correlation rows of every draw need lde x npad doubles of scratch: chunks of draws under the
    // workspace limit and under what the device can give right now (the factor set itself holds most of it)
    size_t limit = h->ws_limit, free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      const size_t margin = size_t(256) << 20;
      const size_t avail = free_b + h->ws_bytes > margin ? free_b + h->ws_bytes - margin : 0;
      if (avail < limit) limit = avail;
    }
    size_t sc = limit / (sizeof(double) * e_stride);
    if (sc < 1) sc = 1;
    if (sc > (size_t)S) sc = S;
    if (sc > 65535) sc = 65535;   // the draw index is a grid dimension
    rc = ensure_ws(h, sizeof(double) * e_stride * sc);
    while (rc == CCGP_ENOMEM && sc > 1) {
      sc = (sc + 1) / 2;
      rc = ensure_ws(h, sizeof(double) * e_stride * sc);
    }
    if (rc) return rc;
    double* E = static_cast<double*>(h->ws);
    DrawView dv{fs->params, S, fs->K, d};
    dv.fam = fs->fam;
    for (int s0 = 0; s0 < S; s0 += (int)sc) {
      const int ns = std::min((int)sc, S - s0);
      CCGP_HIP(hipMemsetAsync(E, 0, sizeof(double) * e_stride * ns, h->stream));
      {
        ScopedTimer t(h, CCGP_T_COV);   // rows t = r(x_t)' (Mixed.corr.vec, HX:425-431)
        launch_cov_cross_batched(h->stream, dXt, m, fs->X, n, d, dv, s0, ns, E, e_stride, lde);
      }
      blocked_predict_from_factors(h, fs->w, n, fs->npad, S, s0, ns, E, e_stride, lde, m, fs->status, fs->sigma2, dmean, dvar);
    }
    CCGP_LAUNCH_CHECK();
  }
  CCGP_HIP(hipMemcpyAsync(out_mean, dmean, sizeof(double) * (size_t)S * m, hipMemcpyDeviceToHost, h->stream));
  CCGP_HIP(hipMemcpyAsync(out_var, dvar, sizeof(double) * (size_t)S * m, hipMemcpyDeviceToHost, h->stream));
  CCGP_HIP(hipStreamSynchronize(h->stream));
  return CCGP_OK;
} CCGP_GUARD_END(h)

size_t ccgp_factorset_bytes(const ccgp_factorset* fs) { return fs ? fs->bytes : 0; }

int ccgp_factorset_free(ccgp_handle* h, ccgp_factorset* fs) try {
  if (!fs) return CCGP_OK;
  if (h) {
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
  }
  if (fs->mem) (void)hipFree(fs->mem);
  delete fs;
  return CCGP_OK;
} CCGP_GUARD_END(h)

}  // extern "C"
