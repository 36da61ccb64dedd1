// Register-resident fused evaluator for the plain likelihood at n <= 128 (BASELINE configs 2
// and 3: the hyperprior grids, HX:549-595 / ADV:552-599; and the likelihood term of logpost).
//
// A G x G grid of threads owns ONE matrix 2-D cyclically in registers: thread (ty, tx) holds
// entries (ty + G a, tx + G b), a >= b, plus one block row of right-hand sides (row ty = 0 is
// y', row ty = 1 is 1').  The covariance is generated straight into those registers (never in
// HBM, never in LDS), then eliminated as L' D L'^T: per column, the owning threads publish the
// column through a 2 x (n + G)-word LDS buffer and every thread applies its rank-1 update on
// registers.  The right-hand-side row is updated like any other row, so the forward
// substitutions come for free (same scheme as small.hip, which keeps the matrix in LDS and
// also serves prediction / inverse / gradient).
//   G = 8 : one WAVE per matrix, no s_barrier at all (LDS traffic of one wave is ordered), four
//           matrices per workgroup -- n <= 64 (Heat-Exchanger grid, n = 64)
//   G = 16: one workgroup per matrix, one barrier per column -- 64 < n <= 128
// NB = ceil(n / G) is a template parameter so that every register index is static.
// INV (round 3, G = 16, NE = NB + 1): the extra rows are the IDENTITY -- solve(R) of logpost (HX:454), one matrix per
// call: Metro's sequential caller.  After the sweep row t holds z_t = L'^-1 e_t; Z goes to LDS and
// R^-1 = Z' D^-1 Z is formed by all 256 threads (the in-LDS kernel of small.hip back-substituted one row per
// thread: 208 us per call at n = 64 against ~50 here).
// NE = number of G-row blocks of EXTRA rows: NE = 1 is the plain likelihood (rows y', 1');
// NE = 4 serves prediction (predict.post, HX:655-673): rows 2.. of the extra block are the
// cross-correlations r(x_t)' of a chunk of G NE - 2 test sites, eliminated like every other row,
// so that afterwards they hold L'^-1 r(x_t) and mean / variance are D-weighted dot products.
//
// Bound: neither HBM nor MFMA -- n dependent elimination steps; algorithmic HBM traffic per
// evaluation is the parameter row in (8 P bytes) and 20 bytes out.
#include <algorithm>
#include <type_traits>

#include "ccgp_internal.h"

namespace ccgp {

namespace {

struct RegArgs {
  const double* X;
  const double* y;
  int n, d;
  const double* params;
  int ldp, K;
  int draw0, B;
  double sigma2;
  int mode;
  double tau2;
  double* loglik;
  double* beta;
  int* status;
  // design-batched form (entropy criteria, BSQ:856-877): every evaluation has its OWN design
  // (X + b * x_stride) and all share one parameter row; only log det R_mixed is wanted
  // prediction (NE > 1): test sites, S = leading dimension of the (draw x test site) tables
  const double* Xt;
  int m, S;
  double* mean;
  double* var;
  size_t x_stride;     // 0: one shared design
  int shared_params;   // 1: params is a single row
  double* logdet;      // optional output: sum_k log d_k
  double* Rinv;        // INV = 1: n x n explicit inverse of the (normalised) mixed correlation matrix
  double* grad;        // INV = 2: d loglik / d params, Btot x P column-major (element (b, j) at grad[b + j * Btot])
  int Btot;
  int grid16;          // CCGP_OPT_SMALL_GRID16: 64 < n <= 104 on the 16 x 16 grid (measurements)
  double* fac;         // FAC instantiation: per draw a block of fac_stride doubles that keeps the factor for predict_sites_kernel
  size_t fac_stride;
};

// ---- the factor a prediction keeps (round 5) ------------------------------------------------------------------------
// Rounds 2 - 4 predicted by carrying the cross-correlation rows of a CHUNK of test sites (30 at n <= 64, 62 above) through the
// elimination as extra rows, one workgroup per (draw, chunk): every chunk generated and factorised the draw's matrix again --
// five times for the 150 sites of a Ground-Vibrations test set.  Now the likelihood instantiation (NE = 1, FAC) factorises
// ONCE per draw and writes L' (unit lower), 1 / d, z'_y, z'_1, beta and 1'R^-1 1 to a block in HBM (11 KB at n = 50); beside it
// -- on a second stream, it needs nothing from the factorisation -- site_corr_kernel forms the test sites' correlation vectors
// r(x_t) with lane = test site; site_solve_kernel then does the forward substitution w' = L'^-1 r and predict.post's arithmetic,
// again lane = site, every matrix operand a broadcast LDS read.  Same operations in the same order per (draw, site) as the
// extra-row scheme, hence the same bits.
//   hdr[8]: beta, s11, bad, sw | rd[NPF] | zy[NPF] | z1[NPF] | L'
// L' in HBM: column-major packed (column k = rows k + 1 .. n - 1, contiguous: the elimination writes a column per step,
// coalesced); predict_sites_kernel re-lays it in LDS by row blocks of eight (lrect / ltri below).
__host__ __device__ constexpr int fac_npf(int n) { return (n + 7) / 8 * 8; }
__host__ __device__ constexpr int fac_col(int n, int k) { return k * (n - 1) - k * (k - 1) / 2; }   // first entry (row k + 1) of column k
struct FacLayout {
  int npf, rd, zy, z1, L, head, total;
  __host__ __device__ FacLayout(int n) {
    npf = fac_npf(n);
    rd = 8; zy = rd + npf; z1 = zy + npf; L = z1 + npf; head = L; total = (L + n * (n - 1) / 2 + 7) / 8 * 8;
  }
};

// doubles of LDS per matrix, from the ACTUAL number of components and dimensions (round 3: sized for kMaxK / kMaxD
// before -- 4 KB of th[] per matrix for a 2 x 4 table -- which left the prediction instances one workgroup per CU short)
constexpr int kGradSlots = 28;   // accumulators of one pass of the gradient contraction: QG (1 + KG) <= 27
__host__ __device__ constexpr int kPerMat(int NP, int G, int NE, int K, int d, bool inv = false) {
  return K * NP /*us*/ + K * d /*th*/ + K /*w2*/ + 2 * (NP + G * NE) /*colbuf*/ + NP /*dvec*/ +
         2 * NP /*zb*/ + 8 +
         (inv ? NP * (NP + 2) + 1 /*Z, row stride NP + 2, 16-byte aligned*/ + 4 * kGradSlots /*wave partial sums of the gradient*/
              : (NE > 1 ? K * G * NE /*ut*/ + 3 * G * G * NE /*partial sums*/ : 0));
}
constexpr int kSmallExpTable = CCGP_SMALL_EXP_TABLE ? kExpTableDoubles : 0;

template <int G>
__device__ __forceinline__ void mat_sync() {
  if constexpr (G == 16) {
    __syncthreads();
  } else {
    // one wave per matrix: its LDS operations are processed in order; only the compiler has to
    // be kept from moving the reads above the writes
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
}

// FULL: n == G * NB exactly (the Heat-Exchanger design: n = 64 = 8 x 8): no padding rows, so the per-entry
// validity tests (r < n, c < n) and the index clamps disappear.  They compiled to one divergent branch per matrix
// entry and component (s_and_saveexec / s_cbranch_execz / s_or + hazard s_nops around every exp): ~10 of the
// ~45 instructions an entry costs in a kernel that is instruction-issue bound.
// Waves per SIMD the register allocator is asked to fit (A/B switches; defaults = what measured fastest, profiles/r03/):
// the kernel is bound by the latency of its per-column LDS round trip as much as by VALU issue, so for n <= 64 a
// fourth wave per SIMD (128 VGPRs, at the price of 84 B of scratch per lane in the generation phase) is worth 8.6 %
// on the Heat-Exchanger grid (9.15 -> 8.37 ms, same box); a fifth wave (102 VGPRs) spills the matrix itself: 17.2 ms.
// G = 16 (64 < n <= 128): 4 stays (5: 6.18 -> 8.23 ms on the 2-D grid).  Prediction instances (NE = 4): 3 instead of 2
// (Ground-Vibrations tables 4.52 -> 4.07 ms; 4: 4.83).
#ifndef CCGP_SMALL_OCC_G8
#define CCGP_SMALL_OCC_G8 4
#endif
#ifndef CCGP_SMALL_OCC_G16
#define CCGP_SMALL_OCC_G16 4
#endif
#ifndef CCGP_FAC_WIDE_MAX
#define CCGP_FAC_WIDE_MAX 64
#endif
#ifndef CCGP_SMALL_OCC_PRED
#define CCGP_SMALL_OCC_PRED 3
#endif
// INV: 0 = none, 1 = explicit inverse (solve(R), HX:454), 2 = analytic gradient of the profile-beta log-likelihood
template <int G, int NB, int NE, bool FULL = false, int INV = 0, bool FAC = false>
__global__ __launch_bounds__(256, INV ? 1 : (NE > 1 ? CCGP_SMALL_OCC_PRED : (G == 8 ? (NB > 8 ? 2 : CCGP_SMALL_OCC_G8) : CCGP_SMALL_OCC_G16)))
void small_reg_kernel(RegArgs a) {
  static_assert(!INV || (G == 16 && NE == NB + 1 && !FULL), "the inverse / gradient runs one matrix per workgroup with n identity rows");
  static_assert(!FAC || (NE == 1 && !FULL && !INV), "the factor is kept by the plain likelihood instantiation");
  constexpr int TPM = G * G;       // threads per matrix
  constexpr int MPW = 256 / TPM;   // matrices per workgroup
  constexpr int NP = G * NB;       // padded order
  constexpr int XR = G * NE;       // extra rows (y', 1', then test sites)
  constexpr int MT = XR - 2;       // test sites per chunk
  extern __shared__ __attribute__((aligned(16))) double smem[];
  // the factorisation a prediction waits for is ONE wave per draw running at the latency of its n columns, with the site
  // correlation kernel's waves issuing beside it on the same SIMDs: it goes first
  if constexpr (FAC) __builtin_amdgcn_s_setprio(3);
  const int n = a.n, d = a.d, K = a.K;
  const int PM = kPerMat(NP, G, NE, K, d, INV != 0);
  const int tid = threadIdx.x, sub = tid / TPM, lt = tid % TPM;
  const int ty = lt % G, tx = lt / G;
  int b = a.draw0 + blockIdx.x * MPW + sub;
  const bool valid = b < a.draw0 + a.B;
  if (!valid) b = a.draw0 + a.B - 1;   // keep the wave alive (shared loads, barriers); results discarded

  double* etab = smem;                        // 2^(j/256) for exp_cov, shared by the workgroup
  double* xs = etab + kSmallExpTable;         // d x n, shared by the matrices of this workgroup
  double* mine = xs + d * n + (size_t)sub * PM;
  double* us = mine;
  double* th = us + K * NP;
  double* w2 = th + K * d;
  double* colbuf = w2 + K;                    // [2][NP + XR]
  double* dvec = colbuf + 2 * (NP + XR);
  double* zb = dvec + NP;                     // [2][NP]
  double* ut = zb + 2 * NP + 8;               // [K][XR]   (NE > 1, prediction)
  double* psum = ut + K * XR;                 // [3][XR][G] (NE > 1, prediction)
  double* zmat = zb + 2 * NP + 8;             // [NP][NP + 2] (INV): row t = L'^-1 e_t, columns scaled by d_c^-1/2
  zmat += (zmat - smem) & 1;                  // its rows are read two doubles at a time (ds_read_b128)
  const int t0 = blockIdx.y * MT;             // first test site of this chunk
  double* xt = xs + (size_t)d * n + (size_t)MPW * PM;     // [d][XR], shared by the workgroup (NE > 1)

  const int pb = a.shared_params ? 0 : b;
  if (CCGP_SMALL_EXP_TABLE) exp_table_load(etab, tid, 256);
  if (a.x_stride == 0) {
    for (int e = tid; e < n * d; e += 256) xs[e] = a.X[e];
  } else {
    // per-evaluation designs: each matrix keeps its own copy right behind the shared slot
    xs = xs + (size_t)d * n + (size_t)MPW * PM + (size_t)sub * d * n;     // (never combined with NE > 1)
    for (int e = lt; e < n * d; e += TPM) xs[e] = a.X[(size_t)b * a.x_stride + e];
  }
  for (int e = lt; e < K * d; e += TPM) th[e] = a.params[pb + (size_t)(K + e) * a.ldp];
  if (lt < K) {
    const double w = a.params[pb + (size_t)lt * a.ldp];
    w2[lt] = w * w;
  }
  if constexpr (NE > 1 && !INV) {
    for (int e = tid; e < d * XR; e += 256) {
      const int k = e / XR, r = e % XR, t = t0 + r - 2;
      xt[e] = (r >= 2 && t < a.m) ? a.Xt[t + (size_t)k * a.m] : 0.0;
    }
  }
  __syncthreads();
  for (int e = lt; e < K * n; e += TPM) {
    const int c = e / n, i = e % n;
    double s = 0.0;
    for (int k = 0; k < d; ++k) { const double v = xs[k * n + i]; s += v * v * th[c * d + k]; }
    us[c * NP + i] = s;
  }
  if constexpr (NE > 1 && !INV) {
    for (int e = lt; e < K * XR; e += TPM) {
      const int c = e / XR, r = e % XR;
      double sq = 0.0;
      for (int k = 0; k < d; ++k) { const double v = xt[k * XR + r]; sq += v * v * th[c * d + k]; }
      ut[c * XR + r] = sq;
    }
  }
  __syncthreads();

  double sw = 0.0;
  for (int c = 0; c < K; ++c) sw += w2[c];
  const double cs = a.sigma2 * sw;
  // (p^2 R1 + (1-p)^2 R2) / (p^2 + (1-p)^2) (HX:412) as a multiplication by the reciprocal: one
  // fp64 division costs ~30 instructions and this kernel is instruction-issue bound (<= 1 ulp apart)
  const double post_scale = (a.mode == 1 ? cs : 1.0) / sw;
  const double post_shift = a.mode == 1 ? a.tau2 : 0.0;

  // ---- generate the lower triangle and the right-hand-side row into registers ------------------
  // Per component q and column-block pair, the k-loop keeps the dot products
  // s[a][j] = sum_k (x_rk theta_qk) x_ck of all the thread's entries in registers, so each
  // (q, k) step costs the thread's row coordinates once per pair instead of three LDS reads
  // and a loop iteration per ENTRY (this kernel is instruction-issue bound).
  double M[NB][NB];   // M[a][b], a >= b
  double E[NE][NB];   // extra rows ty + G e: 0 -> y', 1 -> 1', 2.. -> r(x_t)' of this chunk's test sites
#pragma unroll
  for (int bb = 0; bb < NB; ++bb)
#pragma unroll
    for (int aa = bb; aa < NB; ++aa) M[aa][bb] = 0.0;
  int rowi[NB];   // clamped row of every row block (padding rows repeat row n-1; they are overwritten below)
#pragma unroll
  for (int aa = 0; aa < NB; ++aa) rowi[aa] = FULL ? ty + G * aa : min(ty + G * aa, n - 1);
  for (int q = 0; q < K; ++q) {
    const double wq = w2[q];
#pragma unroll
    for (int bb0 = 0; bb0 < NB; bb0 += 2) {
      constexpr int kPair = 2;
      double sdot[NB][kPair];
#pragma unroll
      for (int aa = 0; aa < NB; ++aa) sdot[aa][0] = sdot[aa][1] = 0.0;
      const int c0 = FULL ? tx + G * bb0 : min(tx + G * bb0, n - 1);
      const int c1 = FULL ? tx + G * (bb0 + 1 < NB ? bb0 + 1 : bb0) : min(tx + G * (bb0 + 1), n - 1);
      for (int k = 0; k < d; ++k) {
        // all LDS reads of this dimension first (one round trip), then the arithmetic: left to itself the
        // compiler reuses one register pair for the row coordinates and waits after every single read
        const double* xk = xs + k * n;
        double xrow[NB];
#pragma unroll
        for (int aa = bb0; aa < NB; ++aa) xrow[aa] = xk[rowi[aa]];
        const double tq = th[q * d + k];
        const double xc0 = xk[c0], xc1 = xk[c1];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int aa = bb0; aa < NB; ++aa) {
          const double xr = xrow[aa] * tq;
          sdot[aa][0] = fma(xr, xc0, sdot[aa][0]);
          if (aa >= bb0 + 1 && bb0 + 1 < NB) sdot[aa][1] = fma(xr, xc1, sdot[aa][1]);
        }
      }
#pragma unroll
      for (int j = 0; j < kPair; ++j) {
        const int bb = bb0 + j;
        if (bb >= NB) continue;
        const int c = tx + G * bb;
#pragma unroll
        for (int aa = 0; aa < NB; ++aa) {
          // constant trip count + a test that folds away: with `aa = bb` as the lower bound the unroller left this
          // loop rolled for odd NB >= 9 (bb is only known once the two loops around it are unrolled), and ONE access
          // with a run-time index puts the whole matrix in scratch (1.6 KB per lane at NB = 13)
          if (aa < bb) continue;
          const int r = ty + G * aa;
          if constexpr (FULL) {
            // every (r, c) is a matrix entry; entries above the diagonal of the diagonal blocks (aa == bb, ty < tx)
            // are computed too and zeroed below -- cheaper than a divergent branch
            const double dist = (us[q * NP + r] + us[q * NP + c]) + (-2.0 * sdot[aa][j]);
            M[aa][bb] = fma(wq, exp_small<!FULL>(dist, etab), M[aa][bb]);
            // pin the finished entry here: without the branch the compiler sinks the tail of every exp (ldexp + mix)
            // to the end of the component loop and keeps two temporaries per entry alive until then (580 B of scratch)
            asm volatile("" : "+v"(M[aa][bb]));
          } else if (r < n && c < n && r >= c) {
            const double dist = (us[q * NP + r] + us[q * NP + c]) + (-2.0 * sdot[aa][j]);
            M[aa][bb] = fma(wq, exp_small<!FULL>(dist, etab), M[aa][bb]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int bb = 0; bb < NB; ++bb) {
    const int c = tx + G * bb;
#pragma unroll
    for (int aa = bb; aa < NB; ++aa) {
      const int r = ty + G * aa;
      if (FULL || (r < n && c < n)) M[aa][bb] = (aa > bb || r >= c) ? fma(post_scale, M[aa][bb], post_shift) : 0.0;
      else M[aa][bb] = r == c ? 1.0 : 0.0;   // identity on the padding
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int ridx = ty + G * e;
      double v = 0.0;
      if (FULL || c < n) {
        if (ridx == 0) v = a.y[c];
        else if (ridx == 1) v = 1.0;
        else if (INV) v = (ridx - 2 == c) ? 1.0 : 0.0;   // identity rows: row 2 + t = e_t'
        else if (NE > 1 && t0 + ridx - 2 < a.m) {
          // Mixed.corr.vec (HX:425-431), corr.vec operation order (HX:373): (theta'x^2 - 2 X Theta x) + u_i
          double acc = 0.0;
          for (int q = 0; q < K; ++q) {
            double sd = 0.0;
            for (int k = 0; k < d; ++k) sd = fma(xs[k * n + c] * th[q * d + k], xt[k * XR + ridx], sd);
            const double dist = (ut[q * XR + ridx] - 2.0 * sd) + us[q * NP + c];
            acc = fma(w2[q], exp_small<!FULL>(dist, etab), acc);
          }
          v = acc / sw;
        }
      }
      E[e][bb] = v;
    }
  }

  // ---- L' D L'^T on registers; one column broadcast through LDS per step ------------------------
  int bad = 0, cur = 0;
  // a bare log-determinant (entropy criteria, BSQ:856-877: the reference calls det(), nothing can "fail") keeps 0
  const double ptol = a.logdet ? 0.0 : pivot_tolerance(a.mode, n);
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
#pragma unroll 1
    for (int kk = 0; kk < G; ++kk) {
      const int k = G * kb + kk;
      if (bad || (!FULL && k >= n)) break;
      double* cb = colbuf + cur * (NP + XR);
      if (tx == kk) {
#pragma unroll
        for (int aa = kb; aa < NB; ++aa) cb[ty + G * aa] = M[aa][kb];
#pragma unroll
        for (int e = 0; e < NE; ++e)
          if (!INV || e < kb + 2) cb[NP + ty + G * e] = E[e][kb];
      }
      mat_sync<G>();
      // every thread of the matrix reads the same word; handing it over through v_readfirstlane tells the compiler
      // that it -- and with it `bad`, the loop exit and the buffer toggle -- is wave-uniform: scalar branch and
      // scalar address arithmetic instead of exec-mask bookkeeping per column
      const double piv_v = cb[k];
      const double piv = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(piv_v)),
                                          __builtin_amdgcn_readfirstlane(__double2loint(piv_v)));
      if (!(piv > ptol)) { bad = k + 1; break; }   // uniform over the matrix's threads (pivot_tolerance: ccgp_internal.h)
      // 1 / pivot by v_rcp_f64 + two Newton steps (full precision, ~6 instructions instead of ~30)
      double rinv = __builtin_amdgcn_rcp(piv);
      rinv = fma(fma(-piv, rinv, 1.0), rinv, rinv);
      rinv = fma(fma(-piv, rinv, 1.0), rinv, rinv);
      if (lt == 0) dvec[k] = piv;
      if constexpr (FAC) {
        // column k of L' (the values every thread forms as lc below), into the row-major packed factor: entry (i, k) of row i
        if (valid) {
          double* Lp = a.fac + (size_t)b * a.fac_stride + FacLayout(n).L + fac_col(n, k) - (k + 1);
          for (int i = k + 1 + lt; i < n; i += TPM) Lp[i] = cb[i] * rinv;
        }
      }
      double lc[NB], lr[NB];
#pragma unroll
      for (int bb = kb; bb < NB; ++bb) lc[bb] = cb[tx + G * bb] * rinv;
      if (tx <= kk) lc[kb] = 0.0;   // columns <= k are finished
#pragma unroll
      for (int aa = kb; aa < NB; ++aa) lr[aa] = cb[ty + G * aa];
      // FULL instantiation (n = G NB, the Heat-Exchanger grid): rows <= k of the diagonal block are NOT masked: lr[kb]
      // only reaches M[kb][kb], and there a row r <= k meets either a finished column (protected by the lc mask
      // above) or a column c > k >= r, i.e. an entry above the diagonal that nothing reads.  Three instructions per
      // column less in an issue-bound kernel.  The general instantiation keeps the mask (round-2 advisor): the
      // unnormalised above-diagonal entries could reach Inf on a nearly singular matrix, and Inf * 0 = NaN would
      // then leak into finished entries with status still 0.
      if constexpr (!FULL) {
        if (ty <= kk) lr[kb] = 0.0;   // rows <= k are finished
      }
      // identity rows (INV): row 2 + t = e_t' is still exactly zero left of column t, so at step k only the rows
      // t <= k have anything to subtract: extra-row blocks e > (k + 2) / G are skipped (kb is unrolled: static)
      constexpr int NEmax = NE;
      const int ne_live = INV ? (kb + 2 < NEmax ? kb + 2 : NEmax) : NEmax;
      double le[NE];
#pragma unroll
      for (int e = 0; e < NE; ++e)
        if (e < ne_live) le[e] = cb[NP + ty + G * e];
#pragma unroll
      for (int bb = kb; bb < NB; ++bb) {
#pragma unroll
        for (int aa = bb; aa < NB; ++aa) M[aa][bb] = fma(-lr[aa], lc[bb], M[aa][bb]);
#pragma unroll
        for (int e = 0; e < NE; ++e)
          if (e < ne_live) E[e][bb] = fma(-le[e], lc[bb], E[e][bb]);
      }
      cur ^= 1;
    }
  }
  // z'_y[c] and z'_1[c] sit in threads ty = 0 / 1
  if (ty < 2) {
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) zb[ty * NP + tx + G * bb] = E[0][bb];
  }
  mat_sync<G>();

  // ---- reductions (first wave of the matrix's threads) --------------------------------------------
  if (lt < 64) {
    const double kNaN = __longlong_as_double(0x7ff8000000000000LL);
    double logdet = 0.0, s11 = 0.0, s1y = 0.0, syy = 0.0;
    if (!bad)
      for (int k = lt; k < n; k += 64) {
        const double dk = dvec[k], zy = zb[k], z1 = zb[NP + k];
        logdet += log(dk);
        s11 += z1 * z1 / dk;
        s1y += z1 * zy / dk;
        syy += zy * zy / dk;
      }
    for (int off = 32; off > 0; off >>= 1) {
      logdet += __shfl_xor(logdet, off, 64);
      s11 += __shfl_xor(s11, off, 64);
      s1y += __shfl_xor(s1y, off, 64);
      syy += __shfl_xor(syy, off, 64);
    }
    const double kLog2Pi = 1.8378770664093454835606594728112;
    double beta = 0.0, ll;
    if (a.mode == 0) {
      beta = s1y / s11;
      double q = 0.0;
      if (!bad)
        for (int k = lt; k < n; k += 64) {
          const double r = zb[k] - beta * zb[NP + k];
          q += r * r / dvec[k];
        }
      for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
      ll = -0.5 * (n * kLog2Pi + n * log(cs) + logdet + q / cs);
    } else {
      ll = -0.5 * (n * kLog2Pi + logdet + syy);
    }
    if (bad) { ll = kNaN; beta = kNaN; }
    if (lt == 0) {
      zb[2 * NP] = beta;      // for the prediction epilogue (slack words behind zb)
      zb[2 * NP + 1] = s11;
    }
    if constexpr (FAC) {
      if (valid) {
        const FacLayout fl(n);
        double* F = a.fac + (size_t)b * a.fac_stride;
        if (lt == 0) { F[0] = beta; F[1] = s11; F[2] = bad ? 1.0 : 0.0; F[3] = sw; }
        for (int c = lt; c < n; c += 64) {
          F[fl.rd + c] = bad ? 0.0 : 1.0 / dvec[c];
          F[fl.zy + c] = zb[c];
          F[fl.z1 + c] = zb[NP + c];
        }
      }
    }
    if (lt == 0 && valid && blockIdx.y == 0) {
      if (a.loglik) a.loglik[b] = ll;
      if (a.beta) a.beta[b] = beta;
      if (a.status) a.status[b] = bad;
      if (a.logdet) a.logdet[b] = bad ? kNaN : logdet;
    }
  }
  if constexpr (INV) {
    // R^-1 = Z' D^-1 Z with Z[t][c] = (L'^-1)[c][t] (zero for c < t).  Round 4: the columns are scaled by d_c^-1/2 on the
    // way to LDS, so an entry of R^-1 is a plain dot product of two rows -- two LDS reads and one FMA per term instead of
    // three reads, a multiplication and an FMA -- and the row stride NP + 2 keeps 16-byte alignment: the dot products
    // start at the even column below i (the entries left of the diagonal are exact zeros) and read two terms per
    // ds_read_b128 (conflict-free: consecutive lanes are 2 (NP + 2) dwords = 4 banks mod 64 apart).
    constexpr int ZS = NP + 2;
    typedef double d2v __attribute__((ext_vector_type(2)));
    const double kNaN = __longlong_as_double(0x7ff8000000000000LL);
    mat_sync<G>();
    for (int c = lt; c < NP; c += TPM) dvec[c] = (bad || c >= n) ? 0.0 : sqrt(1.0 / dvec[c]);
    mat_sync<G>();
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int t = ty + G * e - 2;
#pragma unroll
      for (int bb = 0; bb < NB; ++bb) {
        const int c = tx + G * bb;
        if (t >= 0 && t < NP) zmat[t * ZS + c] = (t < n && c < n) ? E[e][bb] * dvec[c] : 0.0;
      }
    }
    mat_sync<G>();
    // rows i and j of the scaled Z: sum over c >= i (i >= j), two terms per step
    auto row_dot = [&](int i, int j) {
      const d2v* zi = reinterpret_cast<const d2v*>(zmat + i * ZS);
      const d2v* zj = reinterpret_cast<const d2v*>(zmat + j * ZS);
      double acc0 = 0.0, acc1 = 0.0;
      for (int h = i >> 1; h < NP / 2; ++h) {
        const d2v u = zi[h], v = zj[h];
        acc0 = fma(u[0], v[0], acc0);
        acc1 = fma(u[1], v[1], acc1);
      }
      return acc0 + acc1;
    };
    if constexpr (INV == 1) {
      if (valid)
        for (int idx = lt; idx < n * n; idx += TPM) {
          const int i = idx % n, j = idx / n;
          if (i < j) continue;
          double acc = row_dot(i, j);
          if (bad) acc = kNaN;
          a.Rinv[i + (size_t)j * n] = acc;
          a.Rinv[j + (size_t)i * n] = acc;
        }
    } else {
      // ---- gradient (build-defined extension; the reference differentiates numerically, HX:493) ------------------
      //   M = (alpha alpha' - Sigma^-1) / 2,  Sigma = cs R,  cs = sigma2 sum w^2,  alpha = Sigma^-1 (y - beta 1)
      //   d loglik / d w_q      =  2 sigma2 w_q   sum_ab M_ab R_q,ab
      //   d loglik / d theta_qk = -  sigma2 w_q^2 sum_ab M_ab (x_ak - x_bk)^2 R_q,ab
      // (the profile-beta term drops out: d loglik / d beta = 0 at beta_hat).  R^-1 entries come from Z as in the
      // inverse; R_q is regenerated from X; every thread takes pairs (a >= b), off-diagonal pairs count twice.
      // Round 4: the pair's R^-1 entry (an n-long dot product: the O(n^3) part) and m = M_ab are formed ONCE per pass
      // over the pairs, and a pass carries QG components x KG dimensions of accumulators -- K = 2, d = 4 (Heat Exchanger)
      // or K = 3, d = 5 is ONE pass; round 3 walked the pairs K ceil(d / 16) times, recomputing R^-1 every time.
      double* alpha = colbuf;                                  // [NP]   (the column buffers are free now)
      double* part = zmat + NP * ZS;                           // [4][kGradSlots] wave partial sums
      const double beta = zb[2 * NP];
      double* resid = zb;                                      // (z_y - beta z_1)_c d_c^-1/2 in place of z_y
      for (int c = lt; c < NP; c += TPM) resid[c] = c < n ? (zb[c] - beta * zb[NP + c]) * dvec[c] : 0.0;
      mat_sync<G>();
      for (int i = lt; i < n; i += TPM) {
        double s = 0.0;
        for (int c = i; c < n; ++c) s = fma(zmat[i * ZS + c], resid[c], s);
        alpha[i] = s / cs;
      }
      mat_sync<G>();
      const int lane64 = lt & 63, wv = lt >> 6;
      auto contract = [&](auto qg_tag, auto kg_tag) {
        constexpr int QG = decltype(qg_tag)::value, KG = decltype(kg_tag)::value;
        static_assert(QG * (1 + KG) <= kGradSlots, "wave partial sums");
        for (int q0 = 0; q0 < K; q0 += QG) {
          for (int k0 = 0; k0 < d; k0 += KG) {
            double gs[QG], hk[QG][KG];
#pragma unroll
            for (int qq = 0; qq < QG; ++qq) {
              gs[qq] = 0.0;
#pragma unroll
              for (int k = 0; k < KG; ++k) hk[qq][k] = 0.0;
            }
            if (!bad)
              for (int idx = lt; idx < n * n; idx += TPM) {
                const int i = idx % n, j = idx / n;
                if (i < j) continue;
                const double m = (i == j ? 0.5 : 1.0) * (alpha[i] * alpha[j] - row_dot(i, j) / cs);
                double df2[KG];
#pragma unroll
                for (int k = 0; k < KG; ++k) {
                  const int kk = k0 + k < d ? k0 + k : d - 1;
                  const double df = xs[kk * n + i] - xs[kk * n + j];
                  df2[k] = df * df;
                }
#pragma unroll
                for (int qq = 0; qq < QG; ++qq) {
                  const int q = q0 + qq;
                  if (q >= K) break;
                  double sd = 0.0;
                  for (int k = 0; k < d; ++k) sd = fma(xs[k * n + i] * th[q * d + k], xs[k * n + j], sd);
                  const double dist = (us[q * NP + i] + us[q * NP + j]) + (-2.0 * sd);
                  const double v = m * exp_small<true>(dist, etab);
                  gs[qq] += v;
#pragma unroll
                  for (int k = 0; k < KG; ++k) hk[qq][k] = fma(v, df2[k], hk[qq][k]);
                }
              }
#pragma unroll
            for (int qq = 0; qq < QG; ++qq) {
              for (int off = 32; off > 0; off >>= 1) {
                gs[qq] += __shfl_xor(gs[qq], off, 64);
#pragma unroll
                for (int k = 0; k < KG; ++k) hk[qq][k] += __shfl_xor(hk[qq][k], off, 64);
              }
              if (lane64 == 0) {
                part[wv * kGradSlots + qq * (1 + KG)] = gs[qq];
#pragma unroll
                for (int k = 0; k < KG; ++k) part[wv * kGradSlots + qq * (1 + KG) + 1 + k] = hk[qq][k];
              }
            }
            mat_sync<G>();
            if (lt < QG * (1 + KG) && valid) {
              const int qq = lt / (1 + KG), slot = lt % (1 + KG), q = q0 + qq;
              const double tot = (part[lt] + part[kGradSlots + lt]) + (part[2 * kGradSlots + lt] + part[3 * kGradSlots + lt]);
              if (q < K) {
                const double wq = a.params[pb + (size_t)q * a.ldp];
                if (slot == 0) {
                  if (k0 == 0) a.grad[b + (size_t)q * a.Btot] = bad ? kNaN : 2.0 * a.sigma2 * wq * tot;
                } else if (k0 + slot - 1 < d) {
                  a.grad[b + (size_t)(K + q * d + k0 + slot - 1) * a.Btot] = bad ? kNaN : -a.sigma2 * wq * wq * tot;
                }
              }
            }
            mat_sync<G>();
          }
        }
      };
      if (d <= 8) contract(std::integral_constant<int, 3>{}, std::integral_constant<int, 8>{});
      else contract(std::integral_constant<int, 1>{}, std::integral_constant<int, 16>{});
    }
    return;
  }
  if constexpr (NE > 1) {
    // mean = beta + (z_y - beta z_1).w ,  var = sigma2 (1 - w.w + (1 - z_1.w)^2 / (z_1.z_1)),
    // dot products weighted by 1/d_k; w' = extra row of the test site (predict.post, HX:667-670)
    mat_sync<G>();
    const double kNaN = __longlong_as_double(0x7ff8000000000000LL);
    double rd[NB];
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) {
      const int c = tx + G * bb;
      rd[bb] = (c < n && !bad) ? 1.0 / dvec[c] : 0.0;
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      double ww = 0.0, z1w = 0.0, zyw = 0.0;
#pragma unroll
      for (int bb = 0; bb < NB; ++bb) {
        const int c = tx + G * bb;
        if (c < n) {
          const double w = E[e][bb], wr = w * rd[bb];
          ww = fma(w, wr, ww);
          z1w = fma(zb[NP + c], wr, z1w);
          zyw = fma(zb[c], wr, zyw);
        }
      }
      const int ridx = ty + G * e;
      psum[(0 * XR + ridx) * G + tx] = ww;
      psum[(1 * XR + ridx) * G + tx] = z1w;
      psum[(2 * XR + ridx) * G + tx] = zyw;
    }
    mat_sync<G>();
    const double beta = zb[2 * NP], s11 = zb[2 * NP + 1];
    for (int ridx = 2 + lt; ridx < XR; ridx += TPM) {
      const int t = t0 + ridx - 2;
      if (t >= a.m || !valid) continue;
      double ww = 0.0, z1w = 0.0, zyw = 0.0;
#pragma unroll
      for (int x = 0; x < G; ++x) {
        ww += psum[(0 * XR + ridx) * G + x];
        z1w += psum[(1 * XR + ridx) * G + x];
        zyw += psum[(2 * XR + ridx) * G + x];
      }
      const double u = 1.0 - z1w;
      double mean = beta + (zyw - beta * z1w);
      double var = a.sigma2 * (1.0 - ww + u * u / s11);
      if (bad) { mean = kNaN; var = kNaN; }
      a.mean[b + (size_t)t * a.S] = mean;
      a.var[b + (size_t)t * a.S] = var;
    }
  }
}

template <int G, int NB, int NE>
size_t reg_lds_bytes(const RegArgs& a) {
  constexpr int MPW = 256 / (G * G);
  return sizeof(double) * (kSmallExpTable + (size_t)a.d * a.n + (size_t)MPW * kPerMat(G * NB, G, NE, a.K, a.d) +
                           (a.x_stride ? (size_t)MPW * a.d * a.n : 0) + (NE > 1 ? (size_t)a.d * G * NE : 0));
}

template <int G, int NB, int NE = 1>
void launch_one(hipStream_t s, const RegArgs& a) {
  constexpr int MPW = 256 / (G * G);
  const size_t lds = reg_lds_bytes<G, NB, NE>(a);
  static unsigned long long attr_mask = 0;
  once_per_device(attr_mask, [] {
    raise_lds_limit((const void*)small_reg_kernel<G, NB, NE, false>, "small_reg_kernel");
    if constexpr (NE == 1) raise_lds_limit((const void*)small_reg_kernel<G, NB, NE, true>, "small_reg_kernel<full>");
  });
  const bool full = NE == 1 && a.n == G * NB && a.x_stride == 0 && !a.fac;
  const int chunks = NE > 1 ? (a.m + (G * NE - 2) - 1) / (G * NE - 2) : 1;
  const int kMaxGrid = 1 << 20;
  RegArgs c = a;
  for (int b0 = 0; b0 < a.B; b0 += kMaxGrid * MPW) {
    c.draw0 = a.draw0 + b0;
    c.B = a.B - b0 < kMaxGrid * MPW ? a.B - b0 : kMaxGrid * MPW;
    if constexpr (NE == 1) {
      if (a.fac) {   // the likelihood instantiation that keeps its factor for predict_sites_kernel
        static unsigned long long fac_mask = 0;
        once_per_device(fac_mask, [] { raise_lds_limit((const void*)small_reg_kernel<G, NB, NE, false, 0, true>, "small_reg_kernel<fac>"); });
        hipLaunchKernelGGL((small_reg_kernel<G, NB, NE, false, 0, true>), dim3((c.B + MPW - 1) / MPW, 1), dim3(256), lds, s, c);
        continue;
      }
    }
    if constexpr (NE == 1) {
      if (full) {
        hipLaunchKernelGGL((small_reg_kernel<G, NB, NE, true>), dim3((c.B + MPW - 1) / MPW, chunks), dim3(256), lds, s, c);
        continue;
      }
    }
    hipLaunchKernelGGL((small_reg_kernel<G, NB, NE, false>), dim3((c.B + MPW - 1) / MPW, chunks), dim3(256), lds, s, c);
  }
}


// ---- prediction from the kept factor: lane = test site ---------------------------------------------------------------
struct SiteArgs {
  const double* fac;       // ns blocks of fac_stride doubles (FacLayout)
  size_t fac_stride;
  const double* X;         // n x d, column-major
  const double* params;    // draws, column-major with leading dimension ldp
  int ldp;
  int n, d, K;
  const double* Xt;        // m x d, column-major
  int m, S, draw0;         // this launch serves draws draw0 .. of the S x m tables; fac / rs are indexed from 0
  double sigma2;
  double* mean;
  double* var;
  double* rs;              // per (draw, 64-site batch) NPF x 64 doubles: the correlation vectors, lane-major
  int nbatch, npf;         // ceil(m / 64); n rounded up to a multiple of 8
};

// r_i(x_t) = Mixed.corr.vec (HX:425-431) in corr.vec's operation order (HX:373: (theta'x^2 - 2 X Theta x) + u_i) for one draw and
// up to four batches of 64 test sites (one wave each): the scaled training coordinates x_ik theta_qk and u_qi = sum_k theta_qk
// x_ik^2 are formed once per workgroup in LDS and read by broadcast; a lane's site coordinates sit in LDS too ([k][lane]).
// Needs nothing from the factorisation: runs beside it.
template <int KC>
__global__ __launch_bounds__(256, 4) void site_corr_kernel(SiteArgs a) {
  constexpr int RB = 8;
  typedef double d2v __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y, nbatch = a.nbatch, npf = a.npf;
  const int batch = blockIdx.x * (nthr >> 6) + wave;
  const int n = a.n, d = a.d, m = a.m, gb = a.draw0 + b;
  double* th = smem;                                 // [KC][d]
  double* w2 = th + (KC * d + 7) / 8 * 8;            // [8]
  double* us = w2 + 8;                               // [KC][npf]
  double* xs = us + KC * npf;                        // [KC][d][npf]
  double* xw = xs + (size_t)KC * d * npf + (size_t)wave * d * 64;   // this wave's test sites, [k][lane]
  for (int e = tid; e < KC * d; e += nthr) th[e] = a.params[gb + (size_t)(KC + e) * a.ldp];
  if (tid < KC) { const double w = a.params[gb + (size_t)tid * a.ldp]; w2[tid] = w * w; }
  const int t = batch * 64 + lane;
  const int tc = t < m ? t : m - 1;                  // lanes beyond m compute a valid site; nobody reads their column
  if (batch < nbatch)
    for (int k = 0; k < d; ++k) xw[k * 64 + lane] = a.Xt[tc + (size_t)k * m];
  __syncthreads();
  for (int e = tid; e < KC * d * npf; e += nthr) {
    const int i = e % npf, qk = e / npf;             // qk = q * d + k
    xs[e] = i < n ? a.X[i + (size_t)(qk % d) * n] * th[qk] : 0.0;
  }
  for (int e = tid; e < KC * n; e += nthr) {
    const int c = e / n, i = e % n;
    double s = 0.0;
    for (int k = 0; k < d; ++k) { const double v = a.X[i + (size_t)k * n]; s += v * v * th[c * d + k]; }
    us[c * npf + i] = s;
  }
  __syncthreads();
  if (batch >= nbatch) return;
  double sw = 0.0;
  for (int c = 0; c < KC; ++c) sw += w2[c];
  double* __restrict__ rs = a.rs + ((size_t)b * nbatch + batch) * npf * 64 + lane;
  const double* xl = xw + lane;
  double ut[KC];
#pragma unroll
  for (int q = 0; q < KC; ++q) {
    double sq = 0.0;
    for (int k = 0; k < d; ++k) { const double v = xl[k * 64]; sq += v * v * th[q * d + k]; }
    ut[q] = sq;
  }
  for (int i0 = 0; i0 < n; i0 += RB) {
    double sd[KC][RB];
#pragma unroll
    for (int q = 0; q < KC; ++q)
#pragma unroll
      for (int j = 0; j < RB; ++j) sd[q][j] = 0.0;
    double xn = xl[0];
    for (int k = 0; k < d; ++k) {
      const double x = xn;
      xn = xl[(k + 1 < d ? k + 1 : k) * 64];         // the next dimension's coordinate is under way during this one's FMAs
#pragma unroll
      for (int q = 0; q < KC; ++q) {
        const d2v* xv = reinterpret_cast<const d2v*>(xs + (size_t)(q * d + k) * npf + i0);
        const d2v v0 = xv[0], v1 = xv[1], v2 = xv[2], v3 = xv[3];
        sd[q][0] = fma(v0[0], x, sd[q][0]); sd[q][1] = fma(v0[1], x, sd[q][1]);
        sd[q][2] = fma(v1[0], x, sd[q][2]); sd[q][3] = fma(v1[1], x, sd[q][3]);
        sd[q][4] = fma(v2[0], x, sd[q][4]); sd[q][5] = fma(v2[1], x, sd[q][5]);
        sd[q][6] = fma(v3[0], x, sd[q][6]); sd[q][7] = fma(v3[1], x, sd[q][7]);
      }
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      if (i0 + j < n) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < KC; ++q) {
          const double dist = (ut[q] - 2.0 * sd[q][j]) + us[q * npf + i0 + j];
          acc = fma(w2[q], exp_small<true>(dist, nullptr), acc);
        }
        rs[(size_t)(i0 + j) * 64] = acc / sw;
      }
    }
  }
}
size_t site_corr_lds_bytes(int n, int d, int K, int waves) {
  const int npf = fac_npf(n);
  return sizeof(double) * ((size_t)(K * d + 7) / 8 * 8 + 8 + (size_t)K * npf + (size_t)K * d * npf + (size_t)waves * d * 64);
}

// LDS image of L' for the blocked forward substitution: row block I (rows 8 I .. 8 I + 7) holds first its rectangle -- columns
// k < 8 I, each column as the block's 8 row entries side by side (one FMA per row and column, eight independent chains, the
// operands of a column in four ds_read_b128) -- then its 8 x 8 triangle row by row (row j: its j entries).
// (block I starts at sum_{J<I} (64 J + 32) = 32 I^2 words)
__host__ __device__ constexpr int lrect(int I, int k, int j) { return 32 * I * I + k * 8 + j; }
__host__ __device__ constexpr int ltri(int I, int j, int c) { return 32 * I * I + 64 * I + j * (j - 1) / 2 + c; }
template <int NPF>
constexpr size_t site_solve_lds_doubles(int n) { return (size_t)FacLayout(n).head + 32 * (NPF / 8) * (NPF / 8) + 8; }

// One workgroup per (draw, group of up to four batches of 64 test sites), one wave per batch.  The draw's factor block comes
// into LDS once per workgroup (one linear coalesced stream; L' re-laid by row blocks) and every factor operand is then a
// broadcast LDS read.  [A first version read them through scalar loads, one SGPR operand per FMA: 153 us per Ground-Vibrations
// set at n = 50 against 181 for the extra-row scheme -- 10 - 37 KB of L' per wave through a scalar cache that a CU's waves share
// is a latency chain, not a stream.]  Fully unrolled so that w' lives in registers with static indices:
// w'_i = r_i - sum_{k<i} L'_ik w'_k, one accumulator per row, ascending k as the rank-1 updates of the extra-row scheme applied
// them; eight rows at a time: their sums over the finished columns are eight independent chains, then the block's own triangle
// row by row.  Then the three D-weighted dot products in that scheme's summation order (per column class x = c mod GS, classes
// in ascending order; GS = 8 for n <= 64, 16 above: the thread grids the extra rows were spread over): the bits of rounds 2 - 4.
template <int NPF>
__global__ __launch_bounds__(256, NPF <= 48 ? 3 : 2) void site_solve_kernel(SiteArgs a) {
  constexpr int NBL = NPF / 8;
  constexpr int GS = NPF <= 64 ? 8 : 16;
  typedef double d2v __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y, nbatch = a.nbatch;
  const int batch = blockIdx.x * (nthr >> 6) + wave;
  const int n = a.n, m = a.m;
  const FacLayout fl(n);
  const double* __restrict__ Fg = a.fac + (size_t)b * a.fac_stride;
  double* F = smem;                                  // the block's head as it is in HBM ...
  double* L = smem + fl.head;                        // ... and L' in row blocks of 8 (lrect / ltri)
  for (int e = tid; e < fl.head; e += nthr) F[e] = Fg[e];
  {
    // L' arrives as one linear, coalesced stream (four loads in flight per thread); a thread finds the column of its
    // element by walking the column starts forward
    const double* __restrict__ Lg = Fg + fl.L;
    const int total = n * (n - 1) / 2;
    int k = 0, cs = 0, len = n - 1;                  // column k holds rows k + 1 .. n - 1 at [cs, cs + len)
    for (int e0 = tid; e0 < total; e0 += 4 * nthr) {
      double v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = e0 + u * nthr < total ? Lg[e0 + u * nthr] : 0.0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * nthr;
        if (e < total) {
          while (e >= cs + len) { cs += len; --len; ++k; }
          const int i = k + 1 + (e - cs), I = i >> 3, j = i & 7;
          L[k < 8 * I ? lrect(I, k, j) : ltri(I, j, k - 8 * I)] = v[u];
        }
      }
    }
  }
  __syncthreads();
  if (batch >= nbatch) return;
  const int t = batch * 64 + lane;
  const double* __restrict__ rs = a.rs + ((size_t)b * nbatch + batch) * NPF * 64 + lane;

  double w[NPF];
  // the next row block's correlation entries are requested a block ahead where the registers allow (w alone is 2 NPF of them);
  // the scheduling barriers keep the compiler from hoisting every block's loads to the top (1.1 KB of scratch per lane at NPF = 96)
  constexpr bool AHEAD = NPF <= 56;
  double nxt[8];
  if constexpr (AHEAD) {
#pragma unroll
    for (int j = 0; j < 8; ++j) nxt[j] = j < n ? rs[(size_t)j * 64] : 0.0;
  }
#pragma unroll
  for (int I = 0; I < NBL; ++I) {
    double acc[8];
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (AHEAD) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = nxt[j];
      if (I + 1 < NBL) {
#pragma unroll
        for (int j = 0; j < 8; ++j) nxt[j] = (8 * (I + 1) + j < n) ? rs[(size_t)(8 * (I + 1) + j) * 64] : 0.0;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = (8 * I + j < n) ? rs[(size_t)(8 * I + j) * 64] : 0.0;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (8 * I < n) {
      // column k + 1's four ds_read_b128 are issued before column k's eight FMAs, and no further: left alone the scheduler
      // hoists the reads of dozens of columns and spills w
      d2v v0, v1, v2, v3;
      if (I > 0) {
        const d2v* c = reinterpret_cast<const d2v*>(L + lrect(I, 0, 0));
        v0 = c[0]; v1 = c[1]; v2 = c[2]; v3 = c[3];
      }
#pragma unroll
      for (int k = 0; k < 8 * I; ++k) {
        d2v n0 = v0, n1 = v1, n2 = v2, n3 = v3;
        if (k + 1 < 8 * I) {
          const d2v* c = reinterpret_cast<const d2v*>(L + lrect(I, k + 1, 0));
          n0 = c[0]; n1 = c[1]; n2 = c[2]; n3 = c[3];
        }
        const double wk = -w[k];
        acc[0] = fma(wk, v0[0], acc[0]); acc[1] = fma(wk, v0[1], acc[1]);
        acc[2] = fma(wk, v1[0], acc[2]); acc[3] = fma(wk, v1[1], acc[3]);
        acc[4] = fma(wk, v2[0], acc[4]); acc[5] = fma(wk, v2[1], acc[5]);
        acc[6] = fma(wk, v3[0], acc[6]); acc[7] = fma(wk, v3[1], acc[7]);
        __builtin_amdgcn_sched_barrier(0);
        v0 = n0; v1 = n1; v2 = n2; v3 = n3;
      }
#pragma unroll
      for (int j = 1; j < 8; ++j)
#pragma unroll
        for (int c = 0; c < j; ++c) acc[j] = fma(-acc[c], L[ltri(I, j, c)], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) w[8 * I + j] = (8 * I + j < n) ? acc[j] : 0.0;
  }

  double ww = 0.0, z1w = 0.0, zyw = 0.0;
#pragma unroll
  for (int x = 0; x < GS; ++x) {
    double pw = 0.0, p1 = 0.0, py = 0.0;
#pragma unroll
    for (int c = x; c < NPF; c += GS) {
      if (c < n) {
        const double wc = w[c], wr = wc * F[fl.rd + c];
        pw = fma(wc, wr, pw);
        p1 = fma(F[fl.z1 + c], wr, p1);
        py = fma(F[fl.zy + c], wr, py);
      }
    }
    ww += pw;
    z1w += p1;
    zyw += py;
  }
  if (t >= m) return;
  const double beta = F[0], s11 = F[1];
  const double u = 1.0 - z1w;
  double mean = beta + (zyw - beta * z1w);
  double var = a.sigma2 * (1.0 - ww + u * u / s11);
  if (F[2] != 0.0) mean = var = __longlong_as_double(0x7ff8000000000000LL);
  a.mean[(a.draw0 + b) + (size_t)t * a.S] = mean;
  a.var[(a.draw0 + b) + (size_t)t * a.S] = var;
}

static dim3 site_grid(int nbatch, int ns, int* wpb) {
  *wpb = nbatch < 4 ? nbatch : 4;                                  // waves (site batches) per workgroup
  return dim3((nbatch + *wpb - 1) / *wpb, ns);
}
static void launch_site_corr(hipStream_t s, const SiteArgs& a, int ns) {
  int wpb;
  const dim3 grid = site_grid(a.nbatch, ns, &wpb), block(64 * wpb);
  const size_t lds = site_corr_lds_bytes(a.n, a.d, a.K, wpb);
  static unsigned long long attr_mask = 0;
  once_per_device(attr_mask, [] {
    raise_lds_limit((const void*)site_corr_kernel<1>, "site_corr_kernel");
    raise_lds_limit((const void*)site_corr_kernel<2>, "site_corr_kernel");
    raise_lds_limit((const void*)site_corr_kernel<3>, "site_corr_kernel");
  });
  if (a.K == 1) hipLaunchKernelGGL(site_corr_kernel<1>, grid, block, lds, s, a);
  else if (a.K == 2) hipLaunchKernelGGL(site_corr_kernel<2>, grid, block, lds, s, a);
  else hipLaunchKernelGGL(site_corr_kernel<3>, grid, block, lds, s, a);
}
template <int NPF>
static void launch_site_solve(hipStream_t s, const SiteArgs& a, int ns) {
  int wpb;
  const dim3 grid = site_grid(a.nbatch, ns, &wpb), block(64 * wpb);
  static unsigned long long attr_mask = 0;
  once_per_device(attr_mask, [] { raise_lds_limit((const void*)site_solve_kernel<NPF>, "site_solve_kernel"); });
  hipLaunchKernelGGL(site_solve_kernel<NPF>, grid, block, sizeof(double) * site_solve_lds_doubles<NPF>(a.n), s, a);
}

}  // namespace

// Can predict.post tables of this shape go through the kept-factor scheme (else: the extra-row scheme of rounds 2 - 4)?
bool small_reg_sites_supported(int n, int d, int K) {
  if (n > 104 || K > 3) return false;
  const int nb8 = (n + 7) / 8, npf = fac_npf(n);
  return sizeof(double) * (kSmallExpTable + (size_t)d * n + (size_t)4 * kPerMat(8 * nb8, 8, 1, K, d)) <= (size_t)kLdsBytes - 64 &&
         sizeof(double) * ((size_t)FacLayout(n).head + 32 * (npf / 8) * (npf / 8) + 8) <= (size_t)kLdsBytes / 2 - 64 &&   // two per CU
         site_corr_lds_bytes(n, d, K, 4) <= (size_t)kLdsBytes / 2 - 64;
}
// scratch per draw (bytes): the factor block + the correlation vectors of its ceil(m / 64) site batches
size_t small_reg_sites_scratch(int n, int d, int K, int m) {
  (void)d; (void)K;
  return sizeof(double) * ((size_t)FacLayout(n).total + (size_t)((m + 63) / 64) * fac_npf(n) * 64);
}

constexpr int kPredictNE = 4;   // extra row-blocks of the prediction instances: 30 (G = 8) / 62 (G = 16) sites per chunk

// ---- explicit inverse of ONE draw's matrix (solve(R), HX:454) on the register-resident scheme ------------------
template <int NB>
static size_t inv_lds_bytes(int n, int d, int K) {
  return sizeof(double) * (kSmallExpTable + (size_t)d * n + (size_t)kPerMat(16 * NB, 16, NB + 1, K, d, true));
}

bool small_reg_inverse_supported(int n, int d, int K) {
  if (n > 128) return false;
  const int NB = (n + 15) / 16;
  return sizeof(double) * (kSmallExpTable + (size_t)d * n + (size_t)kPerMat(16 * NB, 16, NB + 1, K, d, true)) <=
         (size_t)kLdsBytes - 64;
}

template <int NB, int INV>
static void launch_inv(hipStream_t s, const RegArgs& a) {
  static unsigned long long attr_mask = 0;
  once_per_device(attr_mask, [] {
    raise_lds_limit((const void*)small_reg_kernel<16, NB, NB + 1, false, INV>, INV == 1 ? "small_reg_kernel<inverse>" : "small_reg_kernel<gradient>");
  });
  const int kMaxGrid = 1 << 20;
  RegArgs c = a;
  for (int b0 = 0; b0 < a.B; b0 += kMaxGrid) {
    c.draw0 = a.draw0 + b0;
    c.B = a.B - b0 < kMaxGrid ? a.B - b0 : kMaxGrid;
    hipLaunchKernelGGL((small_reg_kernel<16, NB, NB + 1, false, INV>), dim3(c.B, 1), dim3(256), inv_lds_bytes<NB>(a.n, a.d, a.K), s, c);
  }
}

template <int INV>
static void dispatch_inv(hipStream_t s, const RegArgs& a) {
  switch ((a.n + 15) / 16) {
    case 1: launch_inv<1, INV>(s, a); break;
    case 2: launch_inv<2, INV>(s, a); break;
    case 3: launch_inv<3, INV>(s, a); break;
    case 4: launch_inv<4, INV>(s, a); break;
    case 5: launch_inv<5, INV>(s, a); break;
    case 6: launch_inv<6, INV>(s, a); break;
    case 7: launch_inv<7, INV>(s, a); break;
    default: launch_inv<8, INV>(s, a); break;
  }
}

void launch_small_reg_inverse(hipStream_t s, const double* X, int n, int d, const double* y, DrawView dv, int draw,
                              double sigma2, double* Rinv, double* loglik, double* beta, int* status) {
  RegArgs a{};
  a.X = X; a.y = y; a.n = n; a.d = d; a.params = dv.params; a.ldp = dv.ldp; a.K = dv.K;
  a.draw0 = draw; a.B = 1; a.sigma2 = sigma2; a.mode = 0; a.tau2 = 0.0;
  a.loglik = loglik; a.beta = beta; a.status = status; a.Rinv = Rinv; a.m = 0; a.S = 1;
  dispatch_inv<1>(s, a);
}

// d loglik / d params for B draws (ccgp_loglik_grad_batch, n <= 128): one workgroup per draw on the same scheme
void launch_small_reg_grad(hipStream_t s, const double* X, int n, int d, const double* y, DrawView dv, int B,
                           double sigma2, double* loglik, double* beta, double* grad, int* status) {
  RegArgs a{};
  a.X = X; a.y = y; a.n = n; a.d = d; a.params = dv.params; a.ldp = dv.ldp; a.K = dv.K;
  a.draw0 = 0; a.B = B; a.sigma2 = sigma2; a.mode = 0; a.tau2 = 0.0;
  a.loglik = loglik; a.beta = beta; a.status = status; a.grad = grad; a.Btot = B; a.m = 0; a.S = 1;
  dispatch_inv<2>(s, a);
}

bool small_reg_supported(int n, int d, int K, bool per_design, bool predict) {
  if (n > 128) return false;
  const int G = n <= 64 ? 8 : 16;
  const int NB = (n + G - 1) / G;
  const int MPW = 256 / (G * G);
  const int NE = predict ? kPredictNE : 1;
  return sizeof(double) * (kSmallExpTable + (size_t)d * n + (size_t)MPW * kPerMat(G * NB, G, NE, K, d) +
                           (per_design ? (size_t)MPW * d * n : 0) + (predict ? (size_t)d * G * NE : 0)) <=
         (size_t)kLdsBytes - 64;
}

template <int NE = 1>
static void dispatch(hipStream_t s, const RegArgs& a);

void launch_small_reg_loglik(hipStream_t s, const double* X, int n, int d, const double* y, DrawView dv,
                             int B, double sigma2, int mean_mode, double tau2, double* loglik,
                             double* beta, int* status, bool grid16) {
  RegArgs a{};
  a.X = X; a.y = y; a.n = n; a.d = d; a.params = dv.params; a.ldp = dv.ldp; a.K = dv.K;
  a.draw0 = 0; a.B = B; a.sigma2 = sigma2; a.mode = mean_mode; a.tau2 = tau2;
  a.loglik = loglik; a.beta = beta; a.status = status; a.grid16 = grid16;
  dispatch(s, a);
}

// log det R_mixed for B candidate designs (Xs = B blocks of n x d, column-major each) under ONE
// parameter row: Entropy = -det(R) (BSQ:856-861), Augmented.Mixed.Entropy = -det(R_all)/det(R_old)
// (BSQ:869-877, Schur complement).
void launch_small_reg_logdet_designs(hipStream_t s, const double* Xs, int n, int d, DrawView dv, int B,
                                     double* logdet, int* status) {
  RegArgs a{};
  a.X = Xs; a.y = Xs;   // the right-hand-side row is not used; any n readable doubles will do
  a.n = n; a.d = d; a.params = dv.params; a.ldp = dv.ldp; a.K = dv.K;
  a.draw0 = 0; a.B = B; a.sigma2 = 1.0; a.mode = 0; a.tau2 = 0.0;
  a.status = status; a.x_stride = (size_t)n * d; a.shared_params = 1; a.logdet = logdet;
  dispatch(s, a);
}

// predict.post for S draws x m test sites (mean / var are S x m column-major).  scratch (scratch_bytes, may be null): with at
// least small_reg_sites_scratch() bytes per draw of a chunk the kept-factor scheme runs (one factorisation per draw, then
// lane = test site); otherwise the extra-row scheme.
void launch_small_reg_predict(hipStream_t s, const double* X, int n, int d, const double* y, DrawView dv,
                              int S, const double* Xtest, int m, double sigma2, double* mean, double* var,
                              double* beta, int* status, void* scratch, size_t scratch_bytes, hipStream_t aux,
                              hipEvent_t ev_fork, hipEvent_t ev_join) {
  RegArgs a{};
  a.X = X; a.y = y; a.n = n; a.d = d; a.params = dv.params; a.ldp = dv.ldp; a.K = dv.K;
  a.draw0 = 0; a.B = S; a.sigma2 = sigma2; a.mode = 0; a.tau2 = 0.0;
  a.beta = beta; a.status = status; a.Xt = Xtest; a.m = m; a.S = S; a.mean = mean; a.var = var;
  const size_t per = small_reg_sites_scratch(n, d, dv.K, m);
  if (scratch && small_reg_sites_supported(n, d, dv.K) && scratch_bytes >= per) {
    const FacLayout fl(n);
    const int nbatch = (m + 63) / 64;
    const int chunk = (int)std::min<size_t>((size_t)S, std::min<size_t>(scratch_bytes / per, 32768));
    double* fac = static_cast<double*>(scratch);
    double* rs = fac + (size_t)chunk * fl.total;
    for (int s0 = 0; s0 < S; s0 += chunk) {
      const int ns = std::min(chunk, S - s0);
      SiteArgs sa{fac, (size_t)fl.total, X, dv.params, dv.ldp, n, d, dv.K, Xtest, m, S, s0, sigma2, mean, var, rs, nbatch, fl.npf};
      // the correlation vectors need nothing from the factorisation: on the second stream beside it (the factorisation is ONE
      // wave per draw -- 1000 draws leave three quarters of the wave slots empty -- and runs at the latency of its n columns)
      if (aux && ev_fork && ev_join) {
        (void)hipEventRecord(ev_fork, s);
        (void)hipStreamWaitEvent(aux, ev_fork, 0);
        launch_site_corr(aux, sa, ns);
        (void)hipEventRecord(ev_join, aux);
      } else {
        launch_site_corr(s, sa, ns);
      }
      RegArgs f = a;
      f.Xt = nullptr; f.m = 0; f.mean = f.var = nullptr;
      f.draw0 = s0; f.B = ns;
      f.fac = fac - (size_t)s0 * fl.total;     // the kernel indexes the block by the draw's global number
      f.fac_stride = (size_t)fl.total;
      dispatch<1>(s, f);
      if (aux && ev_fork && ev_join) (void)hipStreamWaitEvent(s, ev_join, 0);
      switch (fl.npf / 8) {
        case 1: launch_site_solve<8>(s, sa, ns); break;
        case 2: launch_site_solve<16>(s, sa, ns); break;
        case 3: launch_site_solve<24>(s, sa, ns); break;
        case 4: launch_site_solve<32>(s, sa, ns); break;
        case 5: launch_site_solve<40>(s, sa, ns); break;
        case 6: launch_site_solve<48>(s, sa, ns); break;
        case 7: launch_site_solve<56>(s, sa, ns); break;
        case 8: launch_site_solve<64>(s, sa, ns); break;
        case 9: launch_site_solve<72>(s, sa, ns); break;
        case 10: launch_site_solve<80>(s, sa, ns); break;
        case 11: launch_site_solve<88>(s, sa, ns); break;
        case 12: launch_site_solve<96>(s, sa, ns); break;
        default: launch_site_solve<104>(s, sa, ns); break;
      }
    }
    return;
  }
  dispatch<kPredictNE>(s, a);
}

template <int NE>
static void dispatch(hipStream_t s, const RegArgs& a) {
  const int n = a.n;
  // A handful of evaluations (Metro's one proposal per logpost call, a speculative batch of a few candidates) is a
  // LATENCY problem: one wave per matrix leaves the chip empty and runs the whole elimination on 64 lanes; the
  // 16 x 16 grid puts four waves on each matrix (n = 64, one evaluation: 41 -> 25 us of kernel time).
  // (with the factor kept -- prediction -- the four-wave form up to 2048 draws: its one factorisation per draw is the critical
  // path of the call and a wave per SIMD is all that 1000 draws fill anyway)
  const bool wide = NE == 1 && a.x_stride == 0 && a.B <= (a.fac ? CCGP_FAC_WIDE_MAX : 64);
  if constexpr (NE == 1) {
    // 64 < n <= 104 (BASELINE config 3: maximin-100): still ONE WAVE per matrix on the 8 x 8 grid, with up to 13 x 13
    // blocks per thread (two waves per SIMD: up to 256 VGPRs) -- at n = 100 1.68 x the minimal FMAs instead of the
    // 2.75 x of the 16 x 16 grid at NB = 7, no s_barrier, and the per-column overhead (pivot, reciprocal, column
    // broadcast) is paid by one wave instead of four: 25.7 k -> ~10 k VALU instructions per evaluation, 6.06 -> 5.07 ms
    // per 103 680 evaluations on the same box, same bits (profiles/r04).  CCGP_OPT_SMALL_GRID16: the 16 x 16 grid (A/B).
    const int nb8 = (n + 7) / 8;
    const bool fits8 = sizeof(double) * (kSmallExpTable + (size_t)a.d * n + (size_t)4 * kPerMat(8 * nb8, 8, 1, a.K, a.d) +
                                         (a.x_stride ? (size_t)4 * a.d * n : 0)) <= (size_t)kLdsBytes - 64;   // four matrices per workgroup
    if (n > 64 && n <= 104 && !wide && fits8 && !a.grid16) {
      switch ((n + 7) / 8) {
        case 9: launch_one<8, 9, NE>(s, a); break;
        case 10: launch_one<8, 10, NE>(s, a); break;
        case 11: launch_one<8, 11, NE>(s, a); break;
        case 12: launch_one<8, 12, NE>(s, a); break;
        default: launch_one<8, 13, NE>(s, a); break;
      }
      return;
    }
  }
  if (n <= 64 && !wide) {
    switch ((n + 7) / 8) {
      case 1: launch_one<8, 1, NE>(s, a); break;
      case 2: launch_one<8, 2, NE>(s, a); break;
      case 3: launch_one<8, 3, NE>(s, a); break;
      case 4: launch_one<8, 4, NE>(s, a); break;
      case 5: launch_one<8, 5, NE>(s, a); break;
      case 6: launch_one<8, 6, NE>(s, a); break;
      case 7: launch_one<8, 7, NE>(s, a); break;
      default: launch_one<8, 8, NE>(s, a); break;
    }
  } else {
    switch ((n + 15) / 16) {
      case 1: if constexpr (NE == 1) { launch_one<16, 1, NE>(s, a); break; }
      case 2: if constexpr (NE == 1) { launch_one<16, 2, NE>(s, a); break; }
      case 3: if constexpr (NE == 1) { launch_one<16, 3, NE>(s, a); break; }
      case 4: if constexpr (NE == 1) { launch_one<16, 4, NE>(s, a); break; }
      case 5: launch_one<16, 5, NE>(s, a); break;
      case 6: launch_one<16, 6, NE>(s, a); break;
      case 7: launch_one<16, 7, NE>(s, a); break;
      default: launch_one<16, 8, NE>(s, a); break;
    }
  }
}

}  // namespace ccgp
