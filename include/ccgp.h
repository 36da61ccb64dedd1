/*
 * libccgp -- C ABI of the MI355X-native Combined-GP evaluation path.
 *
 * The reference (oharari/Convex-Combination-of-Gaussian-Processes) has no FFI: its
 * "API" is a set of global R functions.  Each entry point below replaces the
 * arithmetic of one or more of those functions; the R-side .Call() stub that binds
 * it is in r/ccgp_shim.c and shown in INTEGRATION.md.  Reference abbreviations:
 *   HX  = Heat Exchanger Emulator/Combined GP Heat Exchanger.R
 *   ANI = 2D Codes and Designs/2D Combined GP Anisotropic Public.R
 *   ADV = 2D Codes and Designs/2D Combined GP Isotropic Advanced.R
 *   GV  = Ground Vibrations Emulator/Combined GP Ground Vibrations.R
 *
 * Conventions
 *   - every matrix is fp64, COLUMN-MAJOR (R's native layout), no padding:
 *     X is n x d (X[i + k*n]); a parameter matrix with B draws is B x P
 *     (params[b + j*B]).
 *   - one draw of K component GPs in d dimensions is the row
 *        ( w_1..w_K , theta_{1,1..d} , ... , theta_{K,1..d} ),   P = K + K*d,
 *     and means  Sigma = sigma2 * sum_c w_c^2 R_c(theta_c),
 *                R_c[i,j] = exp(-sum_k theta_ck (x_ik - x_jk)^2).
 *     The reference's (p, theta1, theta2) isotropic draw (HX:408-415) is
 *     w = (p, 1-p), theta_1k = theta1, theta_2k = theta2; its anisotropic draw
 *     (ANI:399-406) is theta_1 = (theta1,theta2), theta_2 = (1+lambda)(theta1,theta2).
 *   - the caller owns every buffer; the library allocates only device scratch kept
 *     in the handle.  A handle is bound to ONE HIP device and is not re-entrant
 *     (R is single-threaded; one handle per host thread / per GPU process); ccgp_multi
 *     (below) puts several devices behind one call.
 *   - return value: 0 ok, <0 error (ccgp_last_error), >0 = number of evaluations in
 *     the batch whose factorisation met a non-positive pivot.  Per-evaluation
 *     status[b] = 0 or 1-based index of the first bad pivot; such an evaluation
 *     returns NaN -- the reference's NA from try(solve(R)) (HX:454-455).
 *   - "_dev" entry points take DEVICE pointers and only enqueue work on the
 *     handle's stream (no allocation, no synchronisation once the workspace has
 *     been sized by ccgp_reserve); the others take HOST pointers and block.
 */
#ifndef CCGP_H
#define CCGP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ccgp_handle ccgp_handle;

enum {
  CCGP_OK = 0,
  CCGP_EINVAL = -1,       /* bad argument */
  CCGP_EHIP = -2,         /* a HIP runtime call failed */
  CCGP_ENOMEM = -3,       /* device workspace allocation failed */
  CCGP_EUNSUPPORTED = -4  /* combination not implemented (message says which) */
};

/* how the mean and the extra variance enter the likelihood */
enum {
  CCGP_MEAN_PROFILE_BETA = 0,  /* logpost: beta = 1'R^-1 y / 1'R^-1 1  (HX:458-460)        */
  CCGP_MEAN_ZERO_PLUS_TAU2 = 1 /* cond.like: mean 0, Sigma + tau^2 11'    (HX:567-570)        */
};

/* which script's log-prior ccgp_logpost adds */
enum {
  CCGP_PRIOR_INVGAMMA = 0, /* HX:462 / ADV:467, pars (a1,b1,a2,b2)                          */
  CCGP_PRIOR_GV = 1,       /* GV:450                                                        */
  CCGP_PRIOR_ISO = 2,      /* ISO:453 = BSQ:450 = D1:636                                    */
  CCGP_PRIOR_ANI = 3       /* ANI:462 (4 transformed parameters, anisotropic kernel)        */
};

/* ---- lifetime --------------------------------------------------------------------- */
int ccgp_create(int device, ccgp_handle** out);
int ccgp_destroy(ccgp_handle* h);
const char* ccgp_last_error(const ccgp_handle* h);
const char* ccgp_version(void);
/* run on a caller-owned hipStream_t (pass NULL to go back to the handle's own stream) */
int ccgp_set_stream(ccgp_handle* h, void* hip_stream);
/* correlation family of every component GP for the calls that follow (default Gaussian).
 * CCGP_KERNEL_GAUSS : R_c[i,j] = exp(-sum_k theta_ck (x_ik - x_jk)^2)     corr.matrix HX:328-337, ANI:351-360
 * CCGP_KERNEL_MATERN: the 1-D scripts' Matern.corr.func(nu, h, theta) D1:348-351,
 *                     R_c[i,j] = z^nu K_nu(z) / (Gamma(nu) 2^(nu-1)),  z = 2 sqrt(nu) |x_i - x_j| / theta_c;
 *                     d must be 1 and a draw is (w_1..w_K, theta_1..theta_K); 1 < nu <= 10, the range the quadrature is validated on (D1:1080 uses 5).
 * Replaces corr.matrix(nu, X, theta) D1:368-374, corr.vec D1:383-389 and everything built on them
 * (Mixed.corr.matrix D1:575-584, logpost D1:609-641, predict.post D1:794-812) through the same entry
 * points as the Gaussian family; the analytic gradient is Gaussian-only.
 * CCGP_KERNEL_MATERN_SPLINE: the two-family 1-D script (1D Combined GP Two Families Public.R = D1F), K = 2:
 *                     component 1 Matern(nu, theta_1) as above, component 2 the non-negative cubic spline
 *                     spline.corr.func(theta_2, h) D1F:346-357.  corr.matrix.combined D1F:453-462 is
 *                     ccgp_mixed_corr_matrix; corr.vec.combined D1F:470-480 is ccgp_mixed_corr_cross and -- as
 *                     written there, the division by p^2 + (1-p)^2 sits after the return -- is NOT normalised;
 *                     predict.post D1F:737-754 (ccgp_predict_batch) uses that same vector. */
enum { CCGP_KERNEL_GAUSS = 0, CCGP_KERNEL_MATERN = 1, CCGP_KERNEL_MATERN_SPLINE = 2 };
int ccgp_set_kernel(ccgp_handle* h, int family, double nu);
/* cap on device scratch used per launch group; larger batches are processed in chunks.  Default: three
 * quarters of the device's memory (216 of 288 GB on MI355X), and never more than is free at call time. */
int ccgp_set_workspace_limit(ccgp_handle* h, size_t bytes);
/* measurement switches (A/B runs under rocprof; results do not depend on them).  Option numbers 0, 1 and 6 were
 * CCGP_OPT_UPDATE_STRIPS, CCGP_OPT_SMALL_LDS and CCGP_OPT_FUSED_COV of rounds 1 - 4 (removed in round 5: no caller outside
 * tests; the fused covariance generation measured 0.3 % and is kept as profiles/r05/fused_cov_removed.diff) and are refused.
 *   CCGP_OPT_FUSE_DIAG      1 (default) = the update launch's diagonal-tile workgroup also factorises and inverts
 *                           the diagonal block; 0 = a separate diag_kernel launch per block column
 *   CCGP_OPT_TAIL_STRIPS    1 (default) = the tiles of an update launch's last, partial step of 256 workgroups run
 *                           as four quarter-width or two half-width strips each, whichever fills the step; 2 = half-width
 *                           strips only (rounds 2 - 3); 0 = every tile whole (same bits all three)
 *   CCGP_OPT_WIDE_OFFSETS   0 (default) = the update loops address their panels through 32-bit buffer offsets wherever a
 *                           panel spans less than 4 GiB; 1 = always the 64-bit-pointer loops that larger matrices
 *                           fall back to (same bits: the tests hold one against the other)
 *   CCGP_OPT_SMALL_GRID16   0 (default) = 64 < n <= 104 runs ONE wave per matrix on the 8 x 8 thread grid (up to 13 x 13
 *                           blocks per thread); 1 = the 16 x 16 grid (one workgroup per matrix) of rounds 1 - 3 (same bits)
 *   CCGP_OPT_PREDICT_FACTOR 1 (default) = prediction tables at n <= 104 (K <= 3) factorise each draw ONCE, keep the factor in HBM
 *                           and solve for the test sites with one lane per site; 0 = the extra-row scheme of rounds 2 - 4 (the
 *                           test sites ride through the elimination in chunks of 30 / 62, one factorisation per chunk): same
 *                           bits, the tests hold one against the other
 *   CCGP_OPT_SCHED          the blocked Cholesky sweep of a chunk (n > 128): 0 = one launch per phase and block column
 *                           (rounds 1 - 4); 1 = ONE persistent launch whose workgroups take diagonal / update / panel-solve
 *                           tiles from dependency-driven queues, two workgroups per CU; 2 = the same with one workgroup per CU;
 *                           3 (default) = 2 for chunks of 32 ... 128 matrices with n >= 2048, where it measures ahead, else 0.
 *                           Same tile code and summation order: same bits.
 *   CCGP_OPT_SCHED_POLICY   bit mask, default 11.  bit 0: a CU's second workgroup only takes a tile while ready tiles are waiting
 *                           (0 = always); bit 1: tiles of a matrix stay on one XCD and synchronise through its L2 (L1
 *                           invalidate + store completion); 0 = agent-scope release / acquire fences, which write back and
 *                           invalidate that L2 per tile (same bits, 40 % slower: DESIGN.md K3); bit 2: keep the per-workgroup
 *                           time account that ccgp_last_sched_profile reads; bit 3: a workgroup goes on with the first task its
 *                           own arrivals made ready instead of queueing it; bit 4 (tests only): drop the announcements of
 *                           matrix 0's second block column, so that the sweep cannot finish -- it must then abort after
 *                           CCGP_SCHED_TIMEOUT_MS (environment, default 30000) and fail every evaluation of the chunk, not hang */
enum { CCGP_OPT_FUSE_DIAG = 2, CCGP_OPT_TAIL_STRIPS = 3, CCGP_OPT_WIDE_OFFSETS = 4, CCGP_OPT_SMALL_GRID16 = 5,
       CCGP_OPT_SCHED = 7, CCGP_OPT_SCHED_POLICY = 8, CCGP_OPT_PREDICT_FACTOR = 9 };
int ccgp_set_option(ccgp_handle* h, int option, int value);
/* pre-size scratch so that later _dev calls of this shape never allocate */
int ccgp_reserve(ccgp_handle* h, int n, int d, int K, int B, int m);
int ccgp_synchronize(ccgp_handle* h);

/* ---- a1/a2/a3: covariance kernels ---------------------------------------------------
 * corr.matrix(X, theta) HX:328-337 / ANI:351-360, corr.matrix.ISO HX:347-356 (caller
 * replicates theta d times); corr.vec / corr.vec.ISO HX:367-375, ANI:369-377 are the
 * m = 1 case of ccgp_corr_cross.  out_R is n x n, out is m x n (row t = r(x_t)). */
int ccgp_corr_matrix(ccgp_handle* h, const double* X, int n, int d, const double* theta,
                     double* out_R);
int ccgp_corr_cross(ccgp_handle* h, const double* Xnew, int m, const double* X, int n, int d,
                    const double* theta, double* out);

/* ---- a4/a5: convex-combination mix ---------------------------------------------------
 * Mixed.corr.matrix HX:408-415, ANI:399-406, ADV:414-421; Mixed.corr.vec HX:425-431,
 * ANI:416-422.  params is ONE row (P doubles). */
int ccgp_mixed_corr_matrix(ccgp_handle* h, const double* X, int n, int d, int K,
                           const double* params, double* out_R);
int ccgp_mixed_corr_cross(ccgp_handle* h, const double* Xnew, int m, const double* X, int n,
                          int d, int K, const double* params, double* out);

/* ---- a6/a7: beta.MLE HX:384-388, sigma2.MLE HX:394-399 (explicit R.Inv given) -------- */
int ccgp_beta_mle(ccgp_handle* h, const double* R_inv, const double* y, int n, double* out_beta);
int ccgp_sigma2_mle(ccgp_handle* h, const double* R_inv, const double* y, int n, double beta,
                    double* out_sigma2);

/* ---- a8/a9/a12: batched log-likelihood ----------------------------------------------
 * For each of B draws: build the mixed covariance, factorise, solve, return
 *   mode 0: dmnorm(y, beta_hat, sigma2*sum(w^2)*R_mixed, log)   (HX:454-460), out_beta
 *   mode 1: dmnorm(y, 0, sigma2*sum(w^2)*R_mixed + tau2 11', log) (HX:567-570), beta = 0
 * out_beta / status may be NULL. */
int ccgp_loglik_batch(ccgp_handle* h, const double* X, int n, int d, const double* y, int K,
                      const double* params, int B, double sigma2, int mean_mode, double tau2,
                      double* out_loglik, double* out_beta, int* status);
int ccgp_loglik_batch_dev(ccgp_handle* h, const double* dX, int n, int d, const double* dy, int K,
                          const double* dparams, int B, double sigma2, int mean_mode, double tau2,
                          double* d_loglik, double* d_beta, int* d_status);

/* build-defined extension (the reference has no analytic gradient; LearnBayes::laplace
 * differences numerically, HX:493): d loglik(mode 0, beta profiled) / d params[b, j],
 * out_grad is B x P column-major.  Any n: n <= 128 in LDS; larger n carries the identity as
 * extra tile rows of the blocked sweep and contracts R^-1 tiles with the kernel derivatives. */
int ccgp_loglik_grad_batch(ccgp_handle* h, const double* X, int n, int d, const double* y, int K,
                           const double* params, int B, double sigma2, double* out_loglik,
                           double* out_beta, double* out_grad, int* status);

/* ---- a8 (+a13): logpost(D.train, theta, y, sigma2[, pars]) -> list(val, beta, R.Inv) --
 * theta_t = (psi1, psi2, phi[, zeta]) on the transformed scale; prior_pars =
 * (a1,b1,a2,b2) for CCGP_PRIOR_INVGAMMA, ignored otherwise.  out_loglik (the bare
 * dmnorm term; ADV:470 returns its exp) and out_Rinv (n x n, solve(R) of HX:454; any n) may
 * be NULL. *status as in the batch call. */
int ccgp_logpost(ccgp_handle* h, const double* X, int n, int d, const double* y, double sigma2,
                 int prior_id, const double* theta_t, const double* prior_pars, double* out_val,
                 double* out_beta, double* out_loglik, double* out_Rinv, int* status);
/* logpost of B transformed parameter vectors in one call -- the candidates of a speculative block of Metro (HX:505-512: the
 * random numbers of an iteration do not depend on accept / reject, so the 2^m - 1 candidates the chain can reach over the next
 * m iterations are known up front; r/ccgp_shim.c: ccgp_R_metro_steps), or a population of starting points.  theta_t is B x q
 * column-major, q = 3 (4 for CCGP_PRIOR_ANI).  val / beta / loglik per row exactly as ccgp_logpost returns them (same device
 * evaluator, same host arithmetic for Jacobian and prior: same bits); NaN and status != 0 where the factorisation fails.
 * Replaces B calls of logpost HX:441-466 (GV:429-454, ISO:433-457, ADV:447-471, ANI:433-467, D1:609-641, D1F:576-602). */
int ccgp_logpost_batch(ccgp_handle* h, const double* X, int n, int d, const double* y, double sigma2, int prior_id,
                       const double* theta_t, int B, const double* prior_pars, double* out_val, double* out_beta,
                       double* out_loglik, int* status);

/* ---- a9: likeli.hyperpars HX:549-575 / choose.hyperpars HX:584-595, ADV:552-599 -------
 * hyper is G x 4 (alpha1 beta1 alpha2 beta2).  For each row: N base-2 Halton quantiles
 * u_j, p = u_j, theta1 = qigamma(u_j, a1, b1), theta2 = qigamma(u_j, a2, b2), mode-1
 * likelihood with tau2 = tau^2, mean of exp() over j (accumulated as log-sum-exp).
 * out[g] = log(mean) if take_log (HX:591) else mean (ADV:595); *out_argmax is the
 * 0-based which.max row.  aniso_lambda < 0: isotropic kernel (HX/ADV); >= 0: the
 * anisotropic kernel of ANI:399-406 with d = 2, (theta1,theta2) per dimension and that
 * fixed lambda (BASELINE config 3).  out_logs (G x N, may be NULL) receives every
 * conditional log-likelihood. */
int ccgp_grid_marginal(ccgp_handle* h, const double* X, int n, int d, const double* y,
                       double sigma2, const double* hyper, int G, int N, double tau, int take_log,
                       double aniso_lambda, double* out, int* out_argmax, double* out_logs);
/* the quadrature nodes the call above uses, exposed for the R surface / tests */
int ccgp_halton_base2(int N, double* out);
int ccgp_qigamma(const double* p, int N, double alpha, double beta, double* out);

/* ---- a10/a11: factors HX:604-613, predict.post HX:655-673 / ANI:604-623 --------------
 * ccgp_predict_batch recomputes, per draw, what Metro caches (R.Inv, beta; HX:515-525)
 * and returns the S x m tables mean[s + t*S], var[s + t*S] that prediction() averages
 * (HX:688-693).  out_beta (S) and status (S) may be NULL.  Any n: n <= 128 runs the fused
 * in-LDS evaluator, larger n appends the m cross-correlation rows to the blocked sweep. */
int ccgp_predict_batch(ccgp_handle* h, const double* X, int n, int d, const double* y, int K,
                       const double* params, int S, const double* Xtest, int m, double sigma2,
                       double* out_mean, double* out_var, double* out_beta, int* status);
int ccgp_predict_batch_dev(ccgp_handle* h, const double* dX, int n, int d, const double* dy, int K,
                           const double* dparams, int S, const double* dXtest, int m,
                           double sigma2, double* d_mean, double* d_var, double* d_beta,
                           int* d_status);
/* ---- device-resident factor set ------------------------------------------------
 * Metro caches R.Inv and beta of every accepted draw (HX:515-525), factors.frame flattens them into a
 * (5 + 2n + n^2)-column data-frame row (HX:625-644) and predict.post re-parses that row for every test
 * site (HX:655-665).  Here the per-draw cache stays on the device as the Cholesky factor the sweep left
 * behind (lower tiles, inverted diagonal blocks, L^-1 y, L^-1 1, beta, 1'R^-1 1): ccgp_factor_batch
 * factorises S draws ONCE, ccgp_predict_from_factorset then serves any number of test sets at the cost
 * of their forward substitutions only (O(m n^2) per draw instead of O(n^3)); results are bit-identical to
 * ccgp_predict_batch.  For n <= 128 the set keeps the draws and the resident inputs only -- the fused
 * evaluator regenerates a factor in registers faster than it could be read back from HBM.
 * All S factors must fit on the device together (142 MB each at n = 4096); CCGP_ENOMEM otherwise.
 * out_loglik / out_beta / status (S each) may be NULL.  The converter that materialises the reference's
 * wide frame for drop-in callers is rsurface.factors_frame_from_draws. */
typedef struct ccgp_factorset ccgp_factorset;
int ccgp_factor_batch(ccgp_handle* h, const double* X, int n, int d, const double* y, int K,
                      const double* params, int S, double sigma2, ccgp_factorset** out,
                      double* out_loglik, double* out_beta, int* status);
/* out_mean / out_var: S x m column-major, as ccgp_predict_batch */
int ccgp_predict_from_factorset(ccgp_handle* h, const ccgp_factorset* fs, const double* Xtest, int m,
                                double* out_mean, double* out_var);
size_t ccgp_factorset_bytes(const ccgp_factorset* fs);
int ccgp_factorset_free(ccgp_handle* h, ccgp_factorset* fs);

/* literal factors(): out = (mean.factor[n], var.factor1[n], var.factor2).  R_inv is taken as the caller gives it, symmetric
 * or not (R's solve() output is symmetric only up to rounding): var.factor1 = apply(R.Inv, 2, sum) are COLUMN sums (HX:609),
 * mean.factor = R.Inv %*% (y - beta) row products (HX:608); ccgp_beta_mle forms (1' R.Inv) y / sum(R.Inv) (HX:387) likewise. */
int ccgp_factors(ccgp_handle* h, const double* R_inv, double beta, const double* y, int n,
                 double* out);
/* literal predict.post arithmetic with caller-supplied cached terms (HX:667-670);
 * r is m x n as produced by ccgp_mixed_corr_cross; out_mean/out_var length m. */
int ccgp_predict_from_factors(ccgp_handle* h, const double* r, int m, int n, double beta,
                              const double* mean_factor, const double* var_factor1,
                              double var_factor2, const double* R_inv, double sigma2,
                              double* out_mean, double* out_var);

/* predict.post(x.new, D.train, pars, sigma2) HX:655-673 as the scripts call it -- once per (draw, test site),
 * HX:688 -- in ONE device round trip: r = Mixed.corr.vec(x_t, D.train, ...) (HX:665, the kernel of
 * ccgp_mixed_corr_cross) for the m rows of Xnew, then the arithmetic of HX:667-670 with the caller's cached
 * terms (the frame row's beta, mean.factor, var.factor1, var.factor2, R.Inv: HX:659-663).  Same bits as
 * ccgp_mixed_corr_cross followed by ccgp_predict_from_factors; X, x.new, the parameter row and the n^2 + 2n cached
 * doubles cross PCIe in one pinned copy, (mean, var) come back in one. */
int ccgp_predict_post(ccgp_handle* h, const double* Xnew, int m, const double* X, int n, int d, int K,
                      const double* params_row, double beta, const double* mean_factor,
                      const double* var_factor1, double var_factor2, const double* R_inv, double sigma2,
                      double* out_mean, double* out_var);

/* ---- entropy criteria of the batch-sequential design script ---------------------------
 * Entropy(D, p, theta1, theta2) = -det(R.mixed(D))  (Batch Sequential ME Design.R:856-861) and
 * Augmented.Mixed.Entropy = -det(R.new - R.cross R.old^-1 R.cross') (same file :869-877), which by
 * the Schur complement equals -det(R.mixed(D.old U D.new)) / det(R.mixed(D.old)): both are log
 * determinants of the mixed correlation matrix of a CANDIDATE DESIGN.  Xs holds B designs of
 * n x d (column-major each, design b at Xs + b*n*d) that share ONE parameter row.  Up to 128 points (the reference's
 * candidate sets are a few dozen) all B designs are evaluated in one launch; larger designs run the blocked sweep one
 * design at a time. */
int ccgp_mixed_logdet_designs(ccgp_handle* h, const double* Xs, int n, int d, int B, int K,
                              const double* params, double* out_logdet, int* status);

/* ---- several GPUs behind ONE host process (csrc/multi.cpp) ------------------------------
 * The drop-in host is R: one single-threaded process.  A ccgp_multi owns one handle per device; the three
 * batched entry points below take the same arguments as their single-device forms, cut the batch into
 * contiguous shards (the grid by grid ROW, so that a row's mean over its Halton nodes stays on one device,
 * HX:574), run the shards concurrently and deliver every result in the caller's host buffers at the
 * shard's offset -- plain device-to-host copies: the consumer (which.max, HX:593-594) is the host, so an
 * all-gather into every GPU's memory would be a copy nobody reads.  Results are bit-identical to the
 * single-device call whatever the number of shards (every evaluation is computed by its own workgroups).
 * devices = NULL means 0 .. n_devices-1; a device may be listed more than once (shards then share it).
 * Return values as for the single-device calls (sum over shards of the failed evaluations; the first
 * negative shard return otherwise, message in ccgp_multi_last_error). */
typedef struct ccgp_multi ccgp_multi;
int ccgp_multi_create(int n_devices, const int* devices, ccgp_multi** out);
int ccgp_multi_destroy(ccgp_multi* m);
int ccgp_multi_count(const ccgp_multi* m);
ccgp_handle* ccgp_multi_handle(ccgp_multi* m, int i); /* per-shard handle: workspace limit, options */
const char* ccgp_multi_last_error(const ccgp_multi* m);
int ccgp_multi_set_kernel(ccgp_multi* m, int family, double nu);
int ccgp_multi_loglik_batch(ccgp_multi* m, const double* X, int n, int d, const double* y, int K,
                            const double* params, int B, double sigma2, int mean_mode, double tau2,
                            double* out_loglik, double* out_beta, int* status);
int ccgp_multi_grid_marginal(ccgp_multi* m, const double* X, int n, int d, const double* y, double sigma2,
                             const double* hyper, int G, int N, double tau, int take_log,
                             double aniso_lambda, double* out, int* out_argmax, double* out_logs);
int ccgp_multi_predict_batch(ccgp_multi* m, const double* X, int n, int d, const double* y, int K,
                             const double* params, int S, const double* Xtest, int mt, double sigma2,
                             double* out_mean, double* out_var, double* out_beta, int* status);

/* ---- measurement hooks (bench.py / rocprof; not part of the R surface) ---------------
 * Time, with HIP events on the handle's stream, every launch group of the most recent
 * *_dev call: ids below.  Returns milliseconds summed over launches and the count. */
enum {
  CCGP_T_COV = 0,      /* covariance / mix build            */
  CCGP_T_UPDATE = 1,   /* blocked Cholesky panel update (MFMA) */
  CCGP_T_DIAG = 2,     /* diagonal-block factor + inverse   */
  CCGP_T_TRSM = 3,     /* panel triangular solve (MFMA)     */
  CCGP_T_SOLVE = 4,    /* forward solves + reductions       */
  CCGP_T_FUSED = 5,    /* small-n fused in-LDS evaluator    */
  CCGP_T_SWEEP = 6,    /* blocked Cholesky sweep as one scheduled launch (update + diag + trsm tiles) */
  CCGP_T_COUNT = 7
};
/* on = 0: off; 1: every id; otherwise a mask with bit (1 + id) set for each id to time */
int ccgp_enable_timing(ccgp_handle* h, int on);
int ccgp_get_timing(ccgp_handle* h, int id, double* out_ms, int* out_launches);
/* With CCGP_OPT_SCHED_POLICY bit 2 set, the scheduled sweep keeps a time account per workgroup (8 words each, ticks of 10 ns:
 * waiting for a task, in diagonal / update / panel-solve tiles, applying arrivals; then tasks run, XCD served, 1 if second on
 * its CU).  Copies the account of the LAST scheduled sweep (synchronises the stream). */
int ccgp_last_sched_profile(ccgp_handle* h, unsigned long long* out, int max_workgroups, int* out_workgroups);

#ifdef __cplusplus
}
#endif
#endif /* CCGP_H */
