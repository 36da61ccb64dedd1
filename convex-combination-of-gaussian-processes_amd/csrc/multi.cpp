// Multi-device entry points of libccgp (include/ccgp.h, "several GPUs behind one host process").
//
// the drop-in host is R -- ONE single-threaded process -- so the 8 GPUs of a node have
// to be reachable from one process through the C ABI, not only through one-process-per-GPU launchers
// (that path is ccgp_amd/shard.py + bench.py: torch.distributed over RCCL).  Every evaluation depends on
// the shared (X, y) and its own parameter row only, so a batch is cut into contiguous shards (the grid by
// grid ROW, so that a row's mean over its Halton nodes stays on one device, HX:574), each shard runs
// through the ordinary single-device entry point on its own handle, and the results land directly in
// the caller's HOST buffers at the shard's offset.
//
// Why plain device-to-host copies and no RCCL all-gather here: the consumer of the gathered vector is the
// host (R's which.max over a REAL() vector, HX:593-594); an all-gather would place a copy of the full
// vector in every GPU's HBM that nobody reads, and then still copy it to the host.  The collective
// belongs to the multi-process path, where every rank needs the full vector (bench.py).
//
// Concurrency: one short-lived host thread per shard for the duration of a call (the single-device entry
// points block, and the grid entry point does its quantile work on the host, which the threads also
// spread over cores).  A handle is touched by exactly one thread at a time; no R API is called here.
#include <algorithm>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ccgp.h"

struct ccgp_multi {
  std::vector<ccgp_handle*> h;
  std::string err;
};

namespace {

// contiguous shard [lo, hi) of `total` items for shard r of w; sizes differ by at most one (shard.py: shard_bounds)
void bounds(int total, int r, int w, int* lo, int* hi) {
  const int base = total / w, rem = total % w;
  *lo = r * base + std::min(r, rem);
  *hi = *lo + base + (r < rem ? 1 : 0);
}

// rows [lo, hi) of a column-major rows x cols matrix, packed
std::vector<double> pack_rows(const double* a, int rows, int cols, int lo, int hi) {
  const int nb = hi - lo;
  std::vector<double> out((size_t)nb * cols);
  for (int j = 0; j < cols; ++j)
    std::copy(a + (size_t)j * rows + lo, a + (size_t)j * rows + hi, out.begin() + (size_t)j * nb);
  return out;
}

constexpr int kShardNoMemory = -1000, kShardException = -1001;   // internal: never returned to the caller

// run fn(shard) on every shard concurrently; collect return codes
template <class F>
int run_shards(ccgp_multi* m, int total, F fn) {
  const int w = (int)m->h.size();
  std::vector<int> rc(w, 0);
  std::vector<std::thread> th;
  for (int r = 0; r < w; ++r) {
    int lo, hi;
    bounds(total, r, w, &lo, &hi);
    if (hi == lo) continue;
    // an exception in the thread body (std::bad_alloc while packing the shard's rows, ...) would otherwise end in
    // std::terminate -- inside R, that is the user's session
    th.emplace_back([&, r, lo, hi] {
      try {
        rc[r] = fn(r, lo, hi);
      } catch (const std::bad_alloc&) {
        rc[r] = kShardNoMemory;
      } catch (...) {
        rc[r] = kShardException;
      }
    });
  }
  for (auto& t : th) t.join();
  int bad = 0;
  for (int r = 0; r < w; ++r) {
    if (rc[r] == kShardNoMemory || rc[r] == kShardException) {
      m->err = "shard " + std::to_string(r) + (rc[r] == kShardNoMemory ? ": host memory exhausted" : ": exception in the shard thread");
      return rc[r] == kShardNoMemory ? CCGP_ENOMEM : CCGP_EHIP;
    }
    if (rc[r] < 0) {
      m->err = "shard " + std::to_string(r) + ": " + ccgp_last_error(m->h[r]);
      return rc[r];
    }
    bad += rc[r];
  }
  return bad;
}

}  // namespace

extern "C" {

int ccgp_multi_create(int n_devices, const int* devices, ccgp_multi** out) {
  if (!out || n_devices < 1 || n_devices > 64) return CCGP_EINVAL;
  *out = nullptr;
  ccgp_multi* m = new ccgp_multi();
  for (int i = 0; i < n_devices; ++i) {
    ccgp_handle* h = nullptr;
    const int rc = ccgp_create(devices ? devices[i] : i, &h);
    if (rc != CCGP_OK) {
      for (ccgp_handle* g : m->h) ccgp_destroy(g);
      delete m;
      return rc;
    }
    m->h.push_back(h);
  }
  *out = m;
  return CCGP_OK;
}

int ccgp_multi_destroy(ccgp_multi* m) {
  if (!m) return CCGP_OK;
  for (ccgp_handle* h : m->h) ccgp_destroy(h);
  delete m;
  return CCGP_OK;
}

int ccgp_multi_count(const ccgp_multi* m) { return m ? (int)m->h.size() : 0; }

ccgp_handle* ccgp_multi_handle(ccgp_multi* m, int i) {
  return (m && i >= 0 && i < (int)m->h.size()) ? m->h[i] : nullptr;
}

const char* ccgp_multi_last_error(const ccgp_multi* m) { return m ? m->err.c_str() : "null handle"; }

int ccgp_multi_set_kernel(ccgp_multi* m, int family, double nu) {
  if (!m) return CCGP_EINVAL;
  for (size_t r = 0; r < m->h.size(); ++r) {
    const int rc = ccgp_set_kernel(m->h[r], family, nu);
    if (rc) {
      m->err = ccgp_last_error(m->h[r]);
      return rc;
    }
  }
  return CCGP_OK;
}

int ccgp_multi_loglik_batch(ccgp_multi* m, const double* X, int n, int d, const double* y, int K,
                            const double* params, int B, double sigma2, int mean_mode, double tau2,
                            double* out_loglik, double* out_beta, int* status) {
  if (!m || !params || !out_loglik || B < 0 || K < 1 || d < 1) return CCGP_EINVAL;
  if (B == 0) return CCGP_OK;
  const int P = K + K * d;
  return run_shards(m, B, [&](int r, int lo, int hi) {
    const std::vector<double> p = pack_rows(params, B, P, lo, hi);
    return ccgp_loglik_batch(m->h[r], X, n, d, y, K, p.data(), hi - lo, sigma2, mean_mode, tau2, out_loglik + lo,
                             out_beta ? out_beta + lo : nullptr, status ? status + lo : nullptr);
  });
}

int ccgp_multi_grid_marginal(ccgp_multi* m, const double* X, int n, int d, const double* y, double sigma2,
                             const double* hyper, int G, int N, double tau, int take_log, double aniso_lambda,
                             double* out, int* out_argmax, double* out_logs) {
  if (!m || !hyper || !out || G < 1 || N < 1) return CCGP_EINVAL;
  const int rc = run_shards(m, G, [&](int r, int lo, int hi) {
    const std::vector<double> hy = pack_rows(hyper, G, 4, lo, hi);
    return ccgp_grid_marginal(m->h[r], X, n, d, y, sigma2, hy.data(), hi - lo, N, tau, take_log, aniso_lambda,
                              out + lo, nullptr, out_logs ? out_logs + (size_t)lo * N : nullptr);
  });
  if (rc < 0) return rc;
  if (out_argmax) {   // which.max over the gathered rows: first maximum, NaN skipped (as ccgp_grid_marginal)
    int best = -1;
    for (int g = 0; g < G; ++g)
      if (out[g] == out[g] && (best < 0 || out[g] > out[best])) best = g;
    *out_argmax = best;
  }
  return rc;
}

int ccgp_multi_predict_batch(ccgp_multi* m, const double* X, int n, int d, const double* y, int K,
                             const double* params, int S, const double* Xtest, int mt, double sigma2,
                             double* out_mean, double* out_var, double* out_beta, int* status) {
  if (!m || !params || !out_mean || !out_var || S < 1 || mt < 1 || K < 1 || d < 1) return CCGP_EINVAL;
  const int P = K + K * d;
  return run_shards(m, S, [&](int r, int lo, int hi) {
    const int nb = hi - lo;
    const std::vector<double> p = pack_rows(params, S, P, lo, hi);
    std::vector<double> mean((size_t)nb * mt), var((size_t)nb * mt);
    const int rc = ccgp_predict_batch(m->h[r], X, n, d, y, K, p.data(), nb, Xtest, mt, sigma2, mean.data(), var.data(),
                                      out_beta ? out_beta + lo : nullptr, status ? status + lo : nullptr);
    if (rc < 0) return rc;
    for (int t = 0; t < mt; ++t) {   // shard tables are nb x m; the caller's are S x m
      std::copy(mean.begin() + (size_t)t * nb, mean.begin() + (size_t)(t + 1) * nb, out_mean + (size_t)t * S + lo);
      std::copy(var.begin() + (size_t)t * nb, var.begin() + (size_t)(t + 1) * nb, out_var + (size_t)t * S + lo);
    }
    return rc;
  });
}

}  // extern "C"
