// Internal declarations shared by the HIP translation units of libccgp (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <atomic>
#include <cstdint>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ccgp.h"
#include "exp_table.h"

namespace ccgp {

constexpr int kTile = 128;       // block size of the blocked Cholesky (rows/cols per tile)
constexpr int kSmallMaxN = 128;  // n <= this goes to the fused in-LDS evaluator
constexpr int kMaxD = 64;        // input dimensions supported by the covariance kernels
constexpr int kMaxK = 8;         // component GPs per draw
constexpr int kLdsBytes = 160 * 1024;

// When does a factorisation "fail" (status != 0, NaN -- the reference's NA)?
//   mean mode 1 (cond.like, HX:561-572): the reference only runs mnormt::dmnorm, whose chol() stops at a
//     NON-POSITIVE pivot -> threshold 0.
//   mean mode 0 (logpost, HX:454-460): the reference first calls solve(R), and base R's solve() refuses a matrix whose
//     reciprocal condition number is below .Machine$double.eps ("system is computationally singular") -> R.Inv <- NA.
//     The device has no condition estimate; it uses a necessary condition that comes for free: the correlation
//     matrix has a unit diagonal and its smallest LDL' pivot d_min bounds the smallest eigenvalue from above, so
//     d_min <= n eps  =>  cond_2(R) >= 1 / (n eps), and the 1-norm condition number LAPACK estimates can be n times
//     cond_2.  A pivot <= n * DBL_EPSILON therefore fails the evaluation.  (A threshold of eps alone is not enough: a
//     duplicated design point leaves a pivot of the size of the rounding error of an n-term sum of squares -- the
//     n = 300 matrix of tests/test_gpu_parity.py::test_non_positive_definite_maps_to_na passed a threshold of eps;
//     before round 3 the threshold was 0 here too, and such a matrix failed or "succeeded" depending on the sign of
//     that rounding error.)
constexpr double kSolvePivotEps = 2.220446049250313e-16;
__host__ __device__ inline double pivot_tolerance(int mean_mode, int n) { return mean_mode == 0 ? n * kSolvePivotEps : 0.0; }

}  // namespace ccgp

namespace ccgp {

// one timed launch group: events are recorded on the handle's stream and only read back
// (hipEventElapsedTime) in ccgp_get_timing, so timing never synchronises the pipeline.
struct TimedSpan {
  int id;
  hipEvent_t e0, e1;
};

// correlation family of the component GPs (ccgp_set_kernel)
struct KernelFamily {
  int id = 0;        // CCGP_KERNEL_GAUSS | CCGP_KERNEL_MATERN | CCGP_KERNEL_MATERN_SPLINE
  double nu = 0.0;   // Matern smoothness
  double norm = 0.0; // 1 / (Gamma(nu) 2^(nu-1))
};

}  // namespace ccgp

namespace ccgp {
constexpr int kPullSlices = 4;
}

struct ccgp_handle {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  size_t ws_limit = size_t(24) << 30;   // ccgp_create replaces this by 3/4 of the device's memory
  int opt_small_grid16 = 0;             // CCGP_OPT_SMALL_GRID16
  int opt_predict_factor = 1;           // CCGP_OPT_PREDICT_FACTOR
  int opt_fuse_diag = 1;                // CCGP_OPT_FUSE_DIAG
  int opt_tail_strips = 1;              // CCGP_OPT_TAIL_STRIPS
  int opt_wide_offsets = 0;             // CCGP_OPT_WIDE_OFFSETS
  int opt_sched = 3;                    // CCGP_OPT_SCHED: 0 = one launch per phase and block column, 1 = dataflow scheduler with two workgroups per CU, 2 = with one, 3 = by chunk size
  int opt_sched_policy = 11;            // CCGP_OPT_SCHED_POLICY bit 0: a CU's second workgroup only takes work while a backlog exists; bit 1: XCD-local synchronisation
  int sched_timeout_ms = 30000;         // a scheduler wait longer than this aborts the sweep (CCGP_SCHED_TIMEOUT_MS)
  int sched_backlog_min = 0;            // 0: one XCD's share of the CUs (CCGP_SCHED_BACKLOG overrides)
  int n_cus = 256;                      // multiProcessorCount of the handle's device
  unsigned long long* sched_prof_dev = nullptr;   // where the last scheduled sweep left its per-workgroup time account (policy bit 2)
  int sched_prof_wgs = 0;
  // grow-only device scratch
  void* ws = nullptr;
  size_t ws_bytes = 0;
  // small staging buffers for the host-pointer entry points
  void* stage = nullptr;
  size_t stage_bytes = 0;
  // pinned host buffer: inputs and results of the one-draw-per-call path (ccgp_logpost) cross PCIe in one copy each
  void* pin = nullptr;
  size_t pin_bytes = 0;
  size_t pin_in = 0;     // bytes of `pin` holding the inputs of the call in flight (results land behind them)
  hipStream_t aux_stream = nullptr;     // second stream of the kept-factor prediction (created on first use)
  hipEvent_t aux_fork = nullptr, aux_join = nullptr;
  hipEvent_t pull_ev[ccgp::kPullSlices] = {};   // one per slice of a large result on its way back (capi.hip: pull)
  std::string err;
  ccgp::KernelFamily fam;
  unsigned timing = 0;   // bit i set: launch groups with id i are bracketed by HIP events
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
  KernelFamily fam;
};

// ---- cov.hip -------------------------------------------------------------------------
// Dense cross / Gram matrix for ONE draw: out[t + i*ldo], t in [0,m) rows of A (m x d),
// i in [0,n) rows of Bm (n x d).  normalise: divide by sum w^2 (Mixed.corr.*).
void launch_cov_dense(hipStream_t s, const double* A, int m, const double* Bm, int n, int d,
                      DrawView dv, int draw, double* out, int ldo);

// Batched lower-triangle tile writer for the blocked path: for draw b0+z, writes
// s*R_mixed + t into the lower tiles of an npad x npad column-major matrix (identity on
// the padding), z in [0, nb).  scale/shift: mode 0 -> (1, 0); mode 1 -> (sigma2*sum w^2, tau2).
// upad (nb x K x npad): u[z][c][i] = sum_k theta_ck x_ik^2 of draw b0 + z, written here once per draw and read by every
// covariance tile of that draw.
void launch_cov_tiles(hipStream_t s, const double* X, int n, int d, DrawView dv, int b0, int nb,
                      double* Abase, size_t batch_stride, int npad, int mean_mode, double sigma2,
                      double tau2, int ld, double* xpad = nullptr, double* upad = nullptr);
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
void launch_small_inverse(hipStream_t s, const double* X, int n, int d, const double* y, DrawView dv, int draw,
                          double sigma2, double* Rinv, double* loglik, double* beta, int* status);
int small_grad_chunks(int n, int d);
// gpart: scratch of B * small_grad_chunks(n, d) * P doubles
void launch_small_grad(hipStream_t s, const double* X, int n, int d, const double* y, DrawView dv,
                       int B, double sigma2, double* loglik, double* beta, double* grad,
                       int* status, double* gpart);

// ---- small_reg.hip: register-resident evaluator for the plain likelihood (n <= 128) ---------
bool small_reg_supported(int n, int d, int K, bool per_design = false, bool predict = false);
void launch_small_reg_predict(hipStream_t s, const double* X, int n, int d, const double* y, DrawView dv,
                              int S, const double* Xtest, int m, double sigma2, double* mean, double* var,
                              double* beta, int* status, void* scratch = nullptr, size_t scratch_bytes = 0,
                              hipStream_t aux = nullptr, hipEvent_t ev_fork = nullptr, hipEvent_t ev_join = nullptr);
bool small_reg_sites_supported(int n, int d, int K);          // kept-factor prediction (round 5): n <= 104, K <= 3
size_t small_reg_sites_scratch(int n, int d, int K, int m);   // bytes of scratch per draw it needs
void launch_small_reg_logdet_designs(hipStream_t s, const double* Xs, int n, int d, DrawView dv, int B,
                                     double* logdet, int* status);
// solve(R) of ONE draw (logpost with R.Inv, HX:454) on the register-resident scheme; likelihood and beta of the same
// factorisation come with it
bool small_reg_inverse_supported(int n, int d, int K);
void launch_small_reg_inverse(hipStream_t s, const double* X, int n, int d, const double* y, DrawView dv, int draw,
                              double sigma2, double* Rinv, double* loglik, double* beta, int* status);
void launch_small_reg_grad(hipStream_t s, const double* X, int n, int d, const double* y, DrawView dv, int B,
                           double sigma2, double* loglik, double* beta, double* grad, int* status);
void launch_small_reg_loglik(hipStream_t s, const double* X, int n, int d, const double* y, DrawView dv,
                             int B, double sigma2, int mean_mode, double tau2, double* loglik,
                             double* beta, int* status, bool grid16 = false);

// ---- blocked.hip ---------------------------------------------------------------------
struct BlockedWs {
  double* A;        // nb matrices of ld x npad (lower tiles + right-hand-side / extra tile rows)
  double* invd;     // nb x nt x 128 x 128 inverses of the diagonal blocks
  double* z;        // nb x nt log-det partials
  double* fin;      // nb x 2: s11 = 1'R^-1 1 and beta per matrix (prediction pass)
  double* xpad;     // npad x kMaxD: the design zero-padded to npad rows (scalar-load source of cov_kernel's columns)
  double* upad;     // nb x kMaxK x npad: u[z][c][i] = sum_k theta_ck x_ik^2 (round 4: shared by cov_kernel and the update)
  double* sched;    // scratch of the dataflow scheduler (sched_ws_bytes): task slots, counters, queue control
  size_t a_stride;  // elements between consecutive matrices
  int ld;           // npad + 128 * (1 + ne)
  int ne;           // extra full tile rows (ceil(m / 128) for prediction, else 0)
};
// optional extra work riding on a blocked sweep (n > 128)
enum { kJobLogdet = -1, kJobPredict = 1, kJobInverse = 2, kJobGrad = 3 };   // >= kJobInverse: identity rows ride along
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
  double* gpart;        // scratch: nb x blocked_grad_partials(npad) x P partial sums
  double* alpha;        // scratch: nb x npad,  R^-1 (y - beta 1)
  // kJobLogdet: log det of the normalised mixed correlation matrix (entropy criteria, BSQ:856-877), indexed like loglik
  double* logdet;
};
bool blocked_grad_supported(int d, int K);
size_t blocked_grad_partials(int npad);   // partial sums per matrix and parameter (gpart = nb x this x P)
size_t blocked_ws_bytes(int npad, int nb, int ne);
size_t sched_ws_bytes(int nt, int nb, int ne);
BlockedWs blocked_carve(void* ws, int npad, int nb, int ne);
// factorise nb matrices in place and finish the likelihood; loglik/beta/status are
// indexed from draw b0.
void blocked_loglik(ccgp_handle* h, const double* X, int n, int d, const double* y, DrawView dv,
                    int b0, int nb, int npad, double sigma2, int mean_mode, double tau2,
                    BlockedWs w, double* loglik, double* beta, int* status,
                    const BlockedJob* job = nullptr);

// forward-substitute the m cross-correlation rows held in E (S blocks of lde x npad, lde = 128 ceil(m / 128),
// row t of block s = r(x_t)' for draw s) through the KEPT factors in w and finish mean / var (S x m)
void blocked_predict_from_factors(ccgp_handle* h, const BlockedWs& w, int n, int npad, int S, int s0, int ns, double* E,
                                  size_t e_stride, int lde, int m, const int* status, double sigma2,
                                  double* mean, double* var);

// ---- special.cpp ---------------------------------------------------------------------
void halton_base2(int N, double* out);
double qigamma_host(double p, double alpha, double beta);   // the __host__ __device__ originals: special_math.h

// ---- helpers -------------------------------------------------------------------------
inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Kernel attributes (dynamic-LDS ceiling) are per device: a process that holds handles on several
// GPUs must set them once on EACH.  `mask` is a function-local static, one bit per device.  Handles of
// different shards (ccgp_multi) call in from different host threads, possibly for the same device: the
// lock is held until the attributes are set, so nobody launches before they are.
inline std::mutex& attr_mutex() {
  static std::mutex m;
  return m;
}
template <class F>
inline void once_per_device(unsigned long long& mask, F set_attributes) {
  std::lock_guard<std::mutex> guard(attr_mutex());
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (mask & bit) return;
  set_attributes();
  mask |= bit;
}

// exp(x) for the covariance kernels (x = -theta-weighted squared distance, x <= 0 up to rounding).
//
// Round 3: table-driven.  exp(x) = 2^m * T[j] * e^r with n = rint(x * 256 / ln 2) = 256 m + j, T[j] = 2^(j/256)
// (256 correctly rounded doubles in LDS, exp_table.h) and |r| <= ln 2 / 512, where a degree-4 polynomial is enough
// (truncation r^5 / 120 <= 4e-17).  n comes out of the "magic number" trick: t = fma(x, 256/ln2, 1.5 * 2^52) holds
// n in the low dword of its mantissa (no v_rndne_f64, no v_cvt_i32_f64) and n as a double is t - 1.5 * 2^52.
// 11 fp64 instructions (max, fma, add, 2 fma for r, 3 fma + mul + fma for T + T (e^r - 1), ldexp) plus 3 cheap
// 32-bit ones and one LDS read, against 17 + 2 for the library routine's degree-11 polynomial that rounds 1 and 2
// used (explicit three-operand v_fma_f64: hipcc's two-address v_fmac_f64 form needs a v_mov_b64 per step) -- in
// kernels that are fp64-VALU-issue bound (PMC: cov_kernel 98 % CU-busy at 0.25 of the HBM peak; 3 x 19 of its 86
// instructions per entry were exp).  Error: <= 1.3 ulp over 2 * 10^5 arguments in [-700, 0] against a 50-digit
// exp (exact-FMA emulation; the library routine: < 1 ulp); tests hold 1e-13 on entries.
// The argument is clamped at -1000 for the index computation only (the magic-number trick needs |n| < 2^31); r is
// formed from the ORIGINAL x, so a NaN distance stays NaN, and v_ldexp_f64 underflows gradually to 0.
static __device__ const unsigned long long kExpTableBits[256] = CCGP_EXP_TABLE_BITS;
constexpr int kExpTableDoubles = 256;

// cooperative copy of the table into LDS (call before a barrier that precedes the first exp_cov)
__device__ __forceinline__ void exp_table_load(double* tab, int tid, int nthreads) {
  for (int e = tid; e < kExpTableDoubles; e += nthreads) tab[e] = __longlong_as_double((long long)kExpTableBits[e]);
}

// exp(-dist): the negation rides on the source modifiers of the first two instructions that read it.  Constants
// sit in SGPRs where the instruction has a single one (gfx9 VOP3: one constant-bus operand), so only 1.5 * 2^52 and
// 1/6 occupy VGPR pairs.
__device__ __forceinline__ double exp_cov(double dist, const double* tab) {
#if !defined(__HIP_DEVICE_COMPILE__)
  (void)tab;
  return exp(-dist);   // host pass of the same translation unit: never called
#else
  const double kScale = __longlong_as_double(0x40771547652b82feLL);     // 256 / ln 2
  const double kNegCHi = __longlong_as_double(0xbf662e42fe000000LL);    // -(ln 2 / 256), 24 trailing zero bits: n * hi is exact
  const double kNegCLo = __longlong_as_double(0xbd9f473de6af278fLL);
  const double kMagic = 6755399441055744.0;                             // 1.5 * 2^52
  const double c4 = __longlong_as_double(0x3fa5555555555555LL), c3 = __longlong_as_double(0x3fc5555555555555LL);
  const double kFloor = -1000.0;
  double xc, t, r, p;
  asm("v_max_f64 %0, -%1, %2" : "=v"(xc) : "v"(dist), "s"(kFloor));    // no canonicalising second v_max
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(t) : "v"(xc), "s"(kScale), "v"(kMagic));
  const double nf = t - kMagic;
  const int n = __double2loint(t);
  asm("v_fma_f64 %0, %1, %2, -%3" : "=v"(r) : "s"(kNegCHi), "v"(nf), "v"(dist));
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "s"(kNegCLo), "v"(nf), "v"(r));
  const double tj = tab[n & 255];
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "v"(r), "s"(c4), "v"(c3));
  asm("v_fma_f64 %0, %1, %2, 0.5" : "=v"(p) : "v"(r), "v"(p));
  asm("v_fma_f64 %0, %1, %2, 1.0" : "=v"(p) : "v"(r), "v"(p));
  const double q = r * p;                                               // e^r - 1
  double v;
  asm("v_fma_f64 %0, %1, %2, %1" : "=v"(v) : "v"(tj), "v"(q));         // T + T (e^r - 1)
  return __builtin_amdgcn_ldexp(v, n >> 8);
#endif
}

// One component's term of one mixed-covariance entry: acc + w_c^2 exp(-((u_row + u_col) - 2 sdot)) (HX:352-356, HX:412), in
// ONE operation order for every producer of a blocked-path entry.
__device__ __forceinline__ double cov_mix_term(double acc, double wc, double u_row, double u_col, double sdot,
                                               const double* tab) {
  const double dist = fma(-2.0, sdot, u_row + u_col);
  return fma(wc, exp_cov(dist, tab), acc);
}

// exp(-dist) WITHOUT a table: the device library's argument reduction and degree-11 polynomial as explicit
// three-operand v_fma_f64 (rounds 1-2; same bits as the library's exp): 17 fp64 instructions + 2.  Kept for the
// register-resident small-n evaluators, where the table variant measured no faster (round 3, profiles/r03/): their
// waves already wait on LDS (column broadcasts of the elimination) and 65 data-dependent table reads per thread add
// bank conflicts to that queue.
// SC: Horner constants as scalar (SGPR) operands instead of VGPRs -- see below
template <bool SC = false>
__device__ __forceinline__ double exp_cov_poly(double dist) {
  const double x = -dist;
#if !defined(__HIP_DEVICE_COMPILE__)
  return exp(x);   // host pass of the same translation unit: never called
#else
  const double kLog2e = __longlong_as_double(0x3ff71547652b82feLL);
  const double kNegLn2Hi = __longlong_as_double(0xbfe62e42fefa39efLL);
  const double kNegLn2Lo = __longlong_as_double(0xbc7abc9e3b39803fLL);
  const double c11 = __longlong_as_double(0x3e5ade156a5dcb37LL), c10 = __longlong_as_double(0x3e928af3fca7ab0cLL),
               c9 = __longlong_as_double(0x3ec71dee623fde64LL), c8 = __longlong_as_double(0x3efa01997c89e6b0LL),
               c7 = __longlong_as_double(0x3f2a01a014761f6eLL), c6 = __longlong_as_double(0x3f56c16c1852b7b0LL),
               c5 = __longlong_as_double(0x3f81111111122322LL), c4 = __longlong_as_double(0x3fa55555555502a1LL),
               c3 = __longlong_as_double(0x3fc5555555555511LL), c2 = __longlong_as_double(0x3fe000000000000bLL);
  const double n = __builtin_rint(x * kLog2e);
  double r = __builtin_fma(kNegLn2Hi, n, x);
  r = __builtin_fma(kNegLn2Lo, n, r);
  double p;
  // Horner constants: "v" operands hold 20 VGPRs, which at four waves per SIMD the compiler re-materialises with ~1.7
  // v_mov_b64 per exp; as "s" operands (a VOP3 instruction takes one scalar operand) they cost SGPRs only.  Which is
  // faster depends on what the register allocator makes of the rest of the kernel (same-box A/B, profiles/r03):
  // scalar constants gain 2 % in the general n = 100 kernel (6.23 -> 6.10 ms) and lose 10 % in the n = 64 FULL
  // instantiation (8.38 -> 9.20 ms: 148 instead of 84 B of scratch), so the caller chooses.
  if constexpr (SC) {
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "v"(r), "s"(c11), "v"(c10));
#define CCGP_FMA3(D, A, B, C) asm("v_fma_f64 %0, %1, %2, %3" : "=v"(D) : "v"(A), "v"(B), "s"(C))
    CCGP_FMA3(p, r, p, c9);
    CCGP_FMA3(p, r, p, c8);
    CCGP_FMA3(p, r, p, c7);
    CCGP_FMA3(p, r, p, c6);
    CCGP_FMA3(p, r, p, c5);
    CCGP_FMA3(p, r, p, c4);
    CCGP_FMA3(p, r, p, c3);
    CCGP_FMA3(p, r, p, c2);
#undef CCGP_FMA3
  } else {
#define CCGP_FMA3(D, A, B, C) asm("v_fma_f64 %0, %1, %2, %3" : "=v"(D) : "v"(A), "v"(B), "v"(C))
    CCGP_FMA3(p, r, c11, c10);
    CCGP_FMA3(p, r, p, c9);
    CCGP_FMA3(p, r, p, c8);
    CCGP_FMA3(p, r, p, c7);
    CCGP_FMA3(p, r, p, c6);
    CCGP_FMA3(p, r, p, c5);
    CCGP_FMA3(p, r, p, c4);
    CCGP_FMA3(p, r, p, c3);
    CCGP_FMA3(p, r, p, c2);
  }
#undef CCGP_FMA3
  p = __builtin_fma(r, p, 1.0);
  p = __builtin_fma(r, p, 1.0);
  return __builtin_amdgcn_ldexp(p, (int)n);
#endif
}

// which exp the small-n evaluators (small_reg.hip, small.hip) use: 0 = polynomial, 1 = table (A/B: -DCCGP_SMALL_EXP_TABLE=1)
#ifndef CCGP_SMALL_EXP_TABLE
#define CCGP_SMALL_EXP_TABLE 0
#endif
template <bool SC = false>
__device__ __forceinline__ double exp_small(double dist, const double* tab) {
#if CCGP_SMALL_EXP_TABLE
  return exp_cov(dist, tab);
#else
  (void)tab;
  return exp_cov_poly<SC>(dist);
#endif
}

// ---- correlation families ---------------------------------------------------------------------------
// Every kernel forms the weighted squared distance  dist = sum_k rate_k (x_ik - x_jk)^2  (in the
// reference's expanded form) and then the correlation.  Gaussian (HX:328-356): rate = theta,
// corr = exp(-dist).  Matern (1-D scripts, D1:348-351): corr = z^nu K_nu(z) / (Gamma(nu) 2^(nu-1)) with
// z = 2 sqrt(nu) |h| / theta, so rate = 4 nu / theta^2 and z = sqrt(dist); d = 1 only, as in the reference.
// Two-family script (D1F:346-357, D1F:453-462): component 1 Matern as above, component 2 the non-negative
// cubic spline  1 - 6u^2 + 6u^3 (u <= 1/2), 2(1-u)^3 (u <= 1), 0 beyond, u = |h| / theta: rate = 1/theta^2.
__device__ __forceinline__ double theta_to_rate(const KernelFamily& f, double theta, int component = 0) {
  if (f.id == 0) return theta;
  if (f.id == 2 && component == 1) return 1.0 / (theta * theta);
  return 4.0 * f.nu / (theta * theta);
}
__device__ __forceinline__ double spline_corr(double u2) {
  const double u = sqrt(u2 > 0.0 ? u2 : 0.0);
  if (u <= 0.5) return 1.0 - 6.0 * u * u + 6.0 * u * u * u;
  if (u <= 1.0) { const double v = 1.0 - u; return 2.0 * v * v * v; }
  return 0.0;
}
// z^nu K_nu(z) / (Gamma(nu) 2^(nu-1)) from  K_nu(z) = int_0^inf exp(-z cosh t) cosh(nu t) dt  by the
// trapezoidal rule (the integrand is entire and decays double-exponentially, so the rule converges
// geometrically): step 0.15 / max(1, sqrt z) gives <= 2e-14 relative error for 1 < nu <= 10 against a
// 40-digit evaluation over z in [1e-6, 300] (tests/test_special.py holds the same rule in numpy); below
// z = 1e-6 the two-term series 1 - z^2 / (4 (nu - 1)).  ~130 terms of three exp each: the Matern family
// is for the reference's small 1-D designs (n = 8), not a throughput path.
__device__ inline double matern_corr(const KernelFamily& f, double z2) {
  if (!(z2 > 1e-12)) return 1.0 - (z2 > 0.0 ? z2 : 0.0) / (4.0 * (f.nu - 1.0));
  const double z = sqrt(z2);
  const double hs = 0.15 / fmax(1.0, sqrt(z));
  double s = 0.5;
  for (int k = 1; k < 6000; ++k) {
    const double t = k * hs;
    const double et = exp(t);
    const double c1 = 0.5 * (et + 1.0 / et) - 1.0;        // cosh t - 1
    // exp(-z (cosh t - 1)) cosh(nu t) with each exponent formed BEFORE exponentiating: exp(nu t) alone
    // overflows for large nu t long before the product does
    const double g = 0.5 * (exp(f.nu * t - z * c1) + exp(-f.nu * t - z * c1));
    s += g;
    if (g < 1e-17 * s && f.nu * t < z * c1) break;
  }
  return exp(f.nu * log(z) - z) * hs * s * f.norm;
}
__device__ __forceinline__ double corr_of_dist(const KernelFamily& f, double dist, const double* tab, int component = 0) {
  if (f.id == 0) return exp_cov(dist, tab);
  if (f.id == 2 && component == 1) return spline_corr(dist);
  return matern_corr(f, dist);
}

// Raise a kernel's dynamic-LDS ceiling to the 160 KiB of a gfx950 CU.  A refusal is remembered PER DEVICE (first one
// wins) and reported by that device's next C-ABI call's launch check instead of surfacing later as an opaque launch
// failure.  raise_lds_limit runs inside once_per_device's lambda, i.e. with attr_mutex held; readers (shard threads of
// ccgp_multi among them) look at an atomic mask first and take the mutex only when their device has an entry.
inline std::string* attr_errors() {
  static std::string e[64];
  return e;
}
inline std::atomic<unsigned long long>& attr_error_mask() {
  static std::atomic<unsigned long long> m{0};
  return m;
}
inline void raise_lds_limit(const void* kernel, const char* name) {
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes - 64);
  if (e == hipSuccess) return;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::string& slot = attr_errors()[dev & 63];
  if (slot.empty()) {
    slot = std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for ") + name + ": " + hipGetErrorString(e);
    attr_error_mask().fetch_or(1ull << (dev & 63));
  }
}
// empty when every attribute request on `device` went through
inline std::string attr_error(int device) {
  if (!(attr_error_mask().load() & (1ull << (device & 63)))) return std::string();
  std::lock_guard<std::mutex> guard(attr_mutex());
  return attr_errors()[device & 63];
}

struct ScopedTimer {
  ccgp_handle* h;
  TimedSpan* sp = nullptr;
  hipStream_t st;
  ScopedTimer(ccgp_handle* hh, int id, hipStream_t stream = nullptr) : h(hh), st(stream ? stream : hh->stream) {
    if (!((h->timing >> id) & 1u)) return;
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
