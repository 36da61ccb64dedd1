// Batched blocked Cholesky + solves for n > 128 (BASELINE config 4: n = 4096).
//
// LEFT-looking, 128-wide block columns, many matrices per launch.  For block column j:
//   update : T_ij = A_ij - sum_{k<j} L_ik L_jk'      all tiles i >= j   (f64 MFMA GEMM)
//   diag   : L_jj = chol(T_jj), W_j = L_jj^-1        one workgroup per matrix, in LDS
//   trsm   : L_ij = T_ij W_j'                        tiles i > j        (f64 MFMA GEMM)
// The right-hand sides [y 1] ride along as an EXTRA (thin) tile row appended below the
// matrix (rows npad .. npad+127 of an (npad+128) x npad array, two of them used): update
// and trsm treat it like any other tile row, so when the sweep ends it holds
// Z' = [y 1]' L^-T, i.e. the forward substitution L z = b is done -- no separate solve
// pass over the 4n^2 B factor.  A last tiny kernel turns Z and the pivots into what the
// reference's dmnorm / beta.MLE return (HX:458-460, HX:570).  Left-looking because every tile is then
// written once and the panels it re-reads are shared through L2 / Infinity Cache: HBM
// traffic is ~2 x 4n^2 B per matrix instead of the right-looking 8n^3/(3 nb) B, so the
// trailing update is MFMA-bound, not HBM-bound.
//
// MFMA: v_mfma_f64_16x16x4_f64.  Operand lane map (one f64 per lane):
//   A[i = lane&15][k = lane>>4],  B[k = lane>>4][j = lane&15],
//   D[i = (lane>>4) + 4*r][j = lane&15] for accumulator register r = 0..3.
// We feed A := Q (the block-column operand) and B := P (the block-row operand), so D's
// lane index runs along matrix ROWS: a wave's store is 4 columns x 128 contiguous bytes.
#include <cstdlib>

#include "ccgp_internal.h"

namespace ccgp {

typedef double d4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kBK = 16;              // k-depth of one LDS stage

struct GemmArgs {
  double* A;
  size_t a_stride;
  int npad;
  double* invd;
  size_t invd_stride;
  int j, nt, nb;
  int ld;    // leading dimension = npad + 128 (right-hand-side tile row) + 128 * ne
  int ne;    // extra full tile rows below the right-hand-side row (cross-correlation rows)
  int mode;  // 0: update, 1: trsm
};

// One output tile strip, C = C - P Q' (MODE 0, update) or C = P Q' (MODE 1, trsm): 128 rows x
// (128 / S) columns per workgroup, K-loop over 16-deep double-buffered LDS stages.
//   S = 1: 2 x 2 waves of 64 x 64      S = 2: 2 x 2 waves of 64 x 32      S = 4: 4 x 1 waves of 32 x 32
// S > 1 exists for wave quantisation: a launch whose workgroup count is 512 m + (a few) would
// leave the chip mostly idle for a whole tile time; cutting every tile into S column strips
// shortens that tail S-fold at the price of re-reading the P panel S times through L2.
// THIN = the right-hand-side tile row: only its first 16 rows carry data, so the waves split
// the strip's columns between them (16 rows x 32 / 16 / 16 columns per wave for S = 1 / 2 / 4).
//
// Staging is LDS-DMA: the next stage is filled by global_load_lds (16 B per lane, one
// wave-instruction = 1 KiB landing linearly in LDS) while the MFMAs of the current stage run,
// so there is no VGPR staging and no ds_write pass (measured 3 % faster than register staging,
// profiles/r01c_strip_selection.md).  Because the DMA destination is lane-linear, the LDS image
// is UNPADDED ([k][128] / [k][CW] doubles) and bank conflicts are removed by an XOR swizzle
// instead: the 16-row block index of column k is XORed with (k & 1), applied on the per-lane
// SOURCE address and again on the fragment read (both sides or neither).
template <int S, bool THIN>
__device__ __forceinline__ void gemm_tile(double* smem, const double* P, int ldP, const double* Q,
                                               int ldQ, int Kdim, double* C, int ld, int mode) {
  constexpr int CW = kTile / S;
  constexpr int NX = THIN ? (S == 1 ? 2 : 1) : (S == 1 ? 4 : 2);
  constexpr int NY = THIN ? 1 : (S == 4 ? 2 : 4);
  constexpr int STAGE = kBK * kTile + kBK * CW;        // doubles per stage, unpadded
  constexpr int CPI = kTile / CW;                      // Q columns covered by one wave-instruction (1, 2, 4)
  constexpr int QI = kBK / CPI / 4;                    // Q wave-instructions per wave per stage (4, 2, 1)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = THIN ? 0 : (S == 4 ? wave * 32 : (wave >> 1) * 64);
  const int col0 = THIN ? wave * NX * 16 : (S == 4 ? 0 : (wave & 1) * (S == 1 ? 64 : 32));
  const bool active = !THIN || col0 < CW;
  const int l15 = lane & 15, l4 = lane >> 4;

  d4 acc[NX][NY];
#pragma unroll
  for (int x = 0; x < NX; ++x)
#pragma unroll
    for (int y = 0; y < NY; ++y) acc[x][y] = d4{0.0, 0.0, 0.0, 0.0};

  // per-lane source rows.  P: wave w issues columns k = w + 4q (parity = w & 1, fixed per wave).
  const int psrc = ((((lane >> 3) ^ (wave & 1)) << 4) + ((lane & 7) << 1));
  const double* pP = P + psrc + (size_t)wave * ldP;
  const size_t stepP = (size_t)4 * ldP;
  // Q: instruction u = w + 4q covers columns CPI*u .. CPI*u + CPI-1; lane -> column CPI*u + lane / (64/CPI)
  constexpr int LPC = 64 / CPI;                        // lanes per Q column
  const int qcol_in = lane / LPC, ql = lane % LPC;
  const int qpar = CPI == 1 ? (wave & 1) : (qcol_in & 1);
  const int qsrc = ((((ql >> 3) ^ qpar) << 4) + ((ql & 7) << 1));
  const double* pQ = Q + qsrc + (size_t)(CPI * wave + qcol_in) * ldQ;
  const size_t stepQ = (size_t)(4 * CPI) * ldQ;

  auto issue = [&](int stage) {
    double* Ps_ = smem + stage * STAGE + wave * kTile;          // wave-uniform LDS bases
    double* Qs_ = smem + stage * STAGE + kBK * kTile + wave * kTile;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pP + q * stepP),
                                       (__attribute__((address_space(3))) void*)(Ps_ + 4 * q * kTile), 16, 0, 0);
#pragma unroll
    for (int q = 0; q < QI; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pQ + q * stepQ),
                                       (__attribute__((address_space(3))) void*)(Qs_ + 4 * q * kTile), 16, 0, 0);
    pP += 4 * stepP;
    pQ += QI * stepQ;
  };

  const int nk = Kdim / kBK;
  issue(0);
  __syncthreads();   // drains the DMA (vmcnt(0)) and publishes it
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) issue((kt + 1) & 1);
    const double* Ps = smem + (kt & 1) * STAGE;
    const double* Qs = Ps + kBK * kTile;
    const int sw = l4 & 1;   // k parity of this lane's fragment element (k = 4 kk + l4)
#pragma unroll
    for (int kk = 0; kk < kBK / 4; ++kk) {
      double pf[NY], qf[NX];
#pragma unroll
      for (int y = 0; y < NY; ++y)
        pf[y] = Ps[(kk * 4 + l4) * kTile + ((((row0 >> 4) + y) ^ sw) << 4) + l15];
#pragma unroll
      for (int x = 0; x < NX; ++x)
        qf[x] = Qs[(kk * 4 + l4) * CW + ((((col0 >> 4) + x) ^ sw) << 4) + l15];
      if (active) {
#pragma unroll
        for (int x = 0; x < NX; ++x)
#pragma unroll
          for (int y = 0; y < NY; ++y)
            acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(qf[x], pf[y], acc[x][y], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  if (!active) return;
#pragma unroll
  for (int x = 0; x < NX; ++x)
#pragma unroll
    for (int y = 0; y < NY; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = col0 + x * 16 + l4 + 4 * r;
        const int row = row0 + y * 16 + l15;
        double* p = C + row + (size_t)col * ld;
        if (mode == 0) *p = *p - acc[x][y][r];
        else *p = acc[x][y][r];
      }
}

template <int S>
constexpr size_t gemm_lds_bytes() {
  return sizeof(double) * 2 * (kBK * kTile + kBK * (kTile / S));   // two unpadded stages
}

template <int MODE, int S>
__device__ __forceinline__ void gemm_dispatch(const GemmArgs& g, double* smem) {
  // tile rows j..nt+ne (update) / j+1..nt+ne (trsm); row nt is the thin right-hand-side tile,
  // rows above nt are full tiles of extra rows (prediction: r(x_t)')
  const int ntile = (MODE == 0 ? g.nt - g.j + 1 : g.nt - g.j) + g.ne;
  // block index -> (matrix, tile, strip): strips of a tile adjacent, tiles of a matrix on one
  // XCD group (blocks are dealt round-robin over the 8 XCDs, so index % 8 labels the group and
  // all tiles of one matrix share the Q panel in that XCD's L2)
  const int L = blockIdx.x;
  const int per_grp = 8 * ntile * S;
  const int grp = L / per_grp, r = L % per_grp;
  const int b = grp * 8 + (r & 7);
  const int t = (r >> 3) / S, strip = (r >> 3) % S;
  if (b >= g.nb) return;
  const int i = g.j + t + (MODE == 0 ? 0 : 1);
  const bool thin = i == g.nt;
  double* Ab = g.A + (size_t)b * g.a_stride;
  const int ld = g.ld;
  const int c0 = strip * (kTile / S);   // first column of this strip inside the tile

  const double* P;
  const double* Q;
  int ldP, ldQ, Kdim;
  if (MODE == 0) {
    P = Ab + (size_t)i * kTile;
    Q = Ab + (size_t)g.j * kTile + c0;
    ldP = ldQ = ld;
    Kdim = g.j * kTile;
  } else {
    P = Ab + (size_t)i * kTile + (size_t)g.j * kTile * ld;
    Q = g.invd + (size_t)b * g.invd_stride + (size_t)g.j * kTile * kTile + c0;
    ldP = ld;
    ldQ = kTile;
    Kdim = kTile;
  }
  double* C = Ab + (size_t)i * kTile + ((size_t)g.j * kTile + c0) * ld;
  if (thin) gemm_tile<S, true>(smem, P, ldP, Q, ldQ, Kdim, C, ld, MODE);
  else gemm_tile<S, false>(smem, P, ldP, Q, ldQ, Kdim, C, ld, MODE);
}

// distinct kernel symbols per phase (rocprof attributes time per symbol) and per strip count
#define CCGP_DEFINE_GEMM(NAME, MODE, S, WPS)                                            \
  __global__ __launch_bounds__(256, WPS) void NAME(GemmArgs g) {                        \
    extern __shared__ __attribute__((aligned(16))) double smem[];                       \
    gemm_dispatch<MODE, S>(g, smem);                                                     \
  }
CCGP_DEFINE_GEMM(chol_update_kernel, 0, 1, 2)
CCGP_DEFINE_GEMM(chol_update_s2_kernel, 0, 2, 2)
CCGP_DEFINE_GEMM(chol_update_s4_kernel, 0, 4, 3)
CCGP_DEFINE_GEMM(chol_trsm_kernel, 1, 1, 2)
CCGP_DEFINE_GEMM(chol_trsm_s2_kernel, 1, 2, 2)
CCGP_DEFINE_GEMM(chol_trsm_s4_kernel, 1, 4, 3)
#undef CCGP_DEFINE_GEMM

// Strip count for an update launch: minimise ceil(workgroups / resident slots) x time per
// workgroup.  Slots per chip (2 / 2 / 3 workgroups per CU by LDS) and the relative per-flop
// efficiency of the narrower strips (1 / 0.85 / 0.70) were fitted to per-launch rocprof
// timings on MI355X (profiles/r01c_strip_selection.md): within 0.5 % of the per-launch optimum.
static int pick_strips(int tiles) {
  const int slots[3] = {512, 512, 768};
  const double eff[3] = {1.0, 0.85, 0.70};
  int best = 1;
  double best_cost = 1e300;
  for (int q = 0; q < 3; ++q) {
    const int S = 1 << q;
    const double cost = (double)((tiles * S + slots[q] - 1) / slots[q]) / (S * eff[q]);
    if (cost < best_cost) { best_cost = cost; best = S; }
  }
  return best;
}

static void launch_gemm(hipStream_t s, const GemmArgs& g, int mode, int S) {
  const int ntile = (mode == 0 ? g.nt - g.j + 1 : g.nt - g.j) + g.ne;
  const int nb8 = round_up(g.nb, 8);
  const dim3 grid(nb8 * ntile * S), block(256);
  if (mode == 0) {
    if (S == 1) hipLaunchKernelGGL(chol_update_kernel, grid, block, gemm_lds_bytes<1>(), s, g);
    else if (S == 2) hipLaunchKernelGGL(chol_update_s2_kernel, grid, block, gemm_lds_bytes<2>(), s, g);
    else hipLaunchKernelGGL(chol_update_s4_kernel, grid, block, gemm_lds_bytes<4>(), s, g);
  } else {
    if (S == 1) hipLaunchKernelGGL(chol_trsm_kernel, grid, block, gemm_lds_bytes<1>(), s, g);
    else if (S == 2) hipLaunchKernelGGL(chol_trsm_s2_kernel, grid, block, gemm_lds_bytes<2>(), s, g);
    else hipLaunchKernelGGL(chol_trsm_s4_kernel, grid, block, gemm_lds_bytes<4>(), s, g);
  }
}

// ---- diagonal block: Cholesky + inverse in LDS ---------------------------------------------
struct DiagArgs {
  double* A;
  size_t a_stride;
  int npad;
  double* invd;
  size_t invd_stride;
  double* logdet_part;  // nb x nt
  int* status;          // indexed from b0
  int j, nt, nb, n;
  int ld;
};

// Register-resident: the 256 threads form a 16 x 16 grid (ty = row class, tx = column class)
// and own the block 2-D cyclically -- thread (ty, tx) holds entries (ty + 16a, tx + 16b).  The
// 128 x 128 block is factorised as L' D L'^T with the identity appended as 128 extra rows, so
// the same rank-1 sweep that eliminates column k also produces L'^-1 (forward substitution of
// e_t): per column one barrier, one 256-word column broadcast through LDS, and <= 44 FMAs per
// thread on registers.  Then L = L' D^1/2 and W = L^-1 = D^-1/2 L'^-1 are written out.
__global__ __launch_bounds__(256) void diag_kernel(DiagArgs g) {
  __shared__ double colbuf[2][256];
  __shared__ double dvec[kTile];
  __shared__ double red[4];
  const int b = blockIdx.x;
  const int tid = threadIdx.x, ty = tid & 15, tx = tid >> 4;
  const int ld = g.ld;
  double* C = g.A + (size_t)b * g.a_stride + (size_t)g.j * kTile + (size_t)g.j * kTile * ld;
  const double kNaN = __longlong_as_double(0x7ff8000000000000LL);

  double M[8][8];  // M[a][b], a >= b: row ty+16a, column tx+16b of the block
  double I[8][8];  // I[a][b], b >= a: appended identity row ty+16a, column tx+16b
#pragma unroll
  for (int bb = 0; bb < 8; ++bb)
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const int r = ty + 16 * a, c = tx + 16 * bb;
      if (a >= bb) M[a][bb] = r >= c ? C[r + (size_t)c * ld] : 0.0;
      if (bb >= a) I[a][bb] = r == c ? 1.0 : 0.0;
    }

  int bad = 0, cur = 0;
#pragma unroll
  for (int kb = 0; kb < 8; ++kb) {
    for (int kk = 0; kk < 16; ++kk) {
      if (bad) break;
      const int k = 16 * kb + kk;
      if (tx == kk) {
#pragma unroll
        for (int a = kb; a < 8; ++a) colbuf[cur][ty + 16 * a] = M[a][kb];
#pragma unroll
        for (int a = 0; a <= kb; ++a) colbuf[cur][kTile + ty + 16 * a] = I[a][kb];
      }
      __syncthreads();
      const double piv = colbuf[cur][k];
      if (!(piv > 0.0)) { bad = k + 1; break; }   // uniform: every thread reads the same word
      const double rinv = 1.0 / piv;
      if (tid == 0) dvec[k] = piv;
      double lc[8], lr[8], li[8];
#pragma unroll
      for (int bb = kb; bb < 8; ++bb) lc[bb] = colbuf[cur][tx + 16 * bb] * rinv;
      if (tx <= kk) lc[kb] = 0.0;   // columns <= k are finished
#pragma unroll
      for (int a = kb; a < 8; ++a) lr[a] = colbuf[cur][ty + 16 * a];
      if (ty <= kk) lr[kb] = 0.0;   // rows <= k are finished
#pragma unroll
      for (int a = 0; a <= kb; ++a) li[a] = colbuf[cur][kTile + ty + 16 * a];
#pragma unroll
      for (int bb = kb; bb < 8; ++bb) {
#pragma unroll
        for (int a = bb; a < 8; ++a) M[a][bb] = fma(-lr[a], lc[bb], M[a][bb]);
#pragma unroll
        for (int a = 0; a <= kb; ++a) I[a][bb] = fma(-li[a], lc[bb], I[a][bb]);
      }
      cur ^= 1;
    }
  }
  __syncthreads();

  double lsum = 0.0;
  if (!bad && tid < kTile) lsum = log(dvec[tid]);
  for (int off = 32; off > 0; off >>= 1) lsum += __shfl_down(lsum, off, 64);
  if ((tid & 63) == 0) red[tid >> 6] = lsum;
  __syncthreads();
  if (tid == 0) {
    g.logdet_part[(size_t)b * g.nt + g.j] = bad ? kNaN : (red[0] + red[1] + red[2] + red[3]);
    if (bad && g.status[b] == 0) g.status[b] = g.j * kTile + bad;
  }

  double* W = g.invd + (size_t)b * g.invd_stride + (size_t)g.j * kTile * kTile;
#pragma unroll
  for (int bb = 0; bb < 8; ++bb) {
    const int c = tx + 16 * bb;
    const double dc = bad ? 1.0 : dvec[c];
    const double sq = sqrt(dc), rs = 1.0 / sq;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const int r = ty + 16 * a;   // block row (L) / identity row index t (W)
      if (a >= bb) {
        double v = r > c ? M[a][bb] * rs : (r == c ? sq : 0.0);
        C[r + (size_t)c * ld] = bad ? kNaN : v;   // poisoned on failure -> NaN likelihood
      }
      // W[c][t] = z'_t[c] / sqrt(d_c), zero above the diagonal (t = r here)
      double w = 0.0;
      if (bb >= a) w = c >= r ? I[a][bb] * rs : 0.0;
      W[c + (size_t)r * kTile] = bad ? kNaN : w;
    }
  }
}

// ---- right-hand-side rows and the final reductions ---------------------------------------------
struct RhsArgs {
  double* A;
  size_t a_stride;
  int npad, n;
  const double* y;
  int ld;
};

// rows npad..npad+127 of every matrix: y' (zero beyond n), 1' (zero beyond n), zeros
__global__ void rhs_rows_kernel(RhsArgs g) {
  const int ld = g.ld;
  double* Ab = g.A + (size_t)blockIdx.y * g.a_stride;
  const int c = blockIdx.x;            // one workgroup of 128 threads per column
  const int r = threadIdx.x;
  double v = 0.0;
  if (c < g.n) v = r == 0 ? g.y[c] : (r == 1 ? 1.0 : 0.0);
  Ab[g.npad + r + (size_t)c * ld] = v;
  // extra tile rows start as zeros (the cross-correlation kernel then fills rows < m, columns < n)
  for (int e = g.npad + kTile + r; e < ld; e += kTile) Ab[e + (size_t)c * ld] = 0.0;
}

struct FinishArgs {
  const double* A;
  size_t a_stride;
  int npad;
  const double* logdet_part;
  const double* params;
  int ldp, K;
  int b0, nt, n;
  double sigma2;
  int mode;
  double* loglik;
  double* beta;
  const int* status;
  int ld;
  double* s11_out;   // nb doubles (1' R^-1 1 in the factor's metric), for the prediction pass
  double* beta_out;  // nb doubles, chunk-local copy of beta
};

__device__ inline double block_sum(double v, double* red, int tid) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void finish_kernel(FinishArgs g) {
  __shared__ double red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int ld = g.ld;
  const double* zrow = g.A + (size_t)b * g.a_stride + g.npad;   // z_y[c] = zrow[c*ld], z_1[c] = zrow[1 + c*ld]
  double logdet = 0.0;
  for (int jb = tid; jb < g.nt; jb += 256) logdet += g.logdet_part[(size_t)b * g.nt + jb];
  logdet = block_sum(logdet, red, tid);
  double s11 = 0.0, s1y = 0.0, syy = 0.0;
  for (int c = tid; c < g.n; c += 256) {
    const double zy = zrow[(size_t)c * ld], z1 = zrow[1 + (size_t)c * ld];
    s11 = fma(z1, z1, s11);
    s1y = fma(z1, zy, s1y);
    syy = fma(zy, zy, syy);
  }
  s11 = block_sum(s11, red, tid);
  s1y = block_sum(s1y, red, tid);
  syy = block_sum(syy, red, tid);
  const double kLog2Pi = 1.8378770664093454835606594728112;
  const int gb = g.b0 + b;
  double sw = 0.0;
  for (int c = 0; c < g.K; ++c) { double w = g.params[gb + (size_t)c * g.ldp]; sw += w * w; }
  const double cs = g.sigma2 * sw;
  double beta = 0.0, ll;
  if (g.mode == 0) {
    beta = s1y / s11;
    double q = 0.0;
    for (int c = tid; c < g.n; c += 256) {
      const double v = zrow[(size_t)c * ld] - beta * zrow[1 + (size_t)c * ld];
      q = fma(v, v, q);
    }
    q = block_sum(q, red, tid);
    ll = -0.5 * (g.n * kLog2Pi + g.n * log(cs) + logdet + q / cs);
  } else {
    ll = -0.5 * (g.n * kLog2Pi + logdet + syy);
  }
  if (tid == 0) {
    const double kNaN = __longlong_as_double(0x7ff8000000000000LL);
    if (g.status && g.status[gb] != 0) { ll = kNaN; beta = kNaN; }
    if (g.loglik) g.loglik[gb] = ll;
    if (g.beta) g.beta[gb] = beta;
    if (g.s11_out) g.s11_out[b] = s11;
    if (g.beta_out) g.beta_out[b] = beta;
  }
}

// ---- prediction from the extra rows: w_t = L^-1 r(x_t) sits in row npad+128+t ---------------------
//   mean = beta + (z_y - beta z_1).w ,  var = sigma2 (1 - w.w + (1 - z_1.w)^2 / (z_1.z_1))
// (predict.post HX:667-670 rewritten in the factor's metric; see small.hip)
struct PredFinishArgs {
  const double* A;
  size_t a_stride;
  int npad, ld, n, m;
  const double* s11;
  const double* beta;
  const int* status;
  int b0, S;
  double sigma2;
  double* mean;
  double* var;
};

__global__ __launch_bounds__(256) void predict_finish_kernel(PredFinishArgs g) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= g.m) return;
  const double* Ab = g.A + (size_t)b * g.a_stride;
  const double* zrow = Ab + g.npad;
  const double* wrow = Ab + g.npad + kTile + t;
  double ww = 0.0, z1w = 0.0, zyw = 0.0;
  for (int c = 0; c < g.n; ++c) {
    const double w = wrow[(size_t)c * g.ld];
    ww = fma(w, w, ww);
    z1w = fma(zrow[1 + (size_t)c * g.ld], w, z1w);
    zyw = fma(zrow[(size_t)c * g.ld], w, zyw);
  }
  const double beta = g.beta[b], s11 = g.s11[b];
  double mean = beta + (zyw - beta * z1w);
  const double u = 1.0 - z1w;
  double var = g.sigma2 * (1.0 - ww + u * u / s11);
  if (g.status && g.status[g.b0 + b] != 0) {
    mean = var = __longlong_as_double(0x7ff8000000000000LL);
  }
  g.mean[(g.b0 + b) + (size_t)t * g.S] = mean;
  g.var[(g.b0 + b) + (size_t)t * g.S] = var;
}

}  // namespace

size_t blocked_ws_bytes(int npad, int nb, int ne) {
  const int nt = npad / kTile;
  size_t dbl = (size_t)nb * (npad + kTile * (1 + ne)) * npad + (size_t)nb * nt * kTile * kTile +
               (size_t)nb * (nt + 2) + 64;
  return dbl * sizeof(double);
}

BlockedWs blocked_carve(void* ws, int npad, int nb, int ne) {
  const int nt = npad / kTile;
  BlockedWs w{};
  w.A = static_cast<double*>(ws);
  w.ld = npad + kTile * (1 + ne);
  w.ne = ne;
  w.a_stride = (size_t)w.ld * npad;
  w.invd = w.A + (size_t)nb * w.a_stride;
  w.z = w.invd + (size_t)nb * nt * kTile * kTile;  // logdet partials (nb x nt), then s11 and beta (nb each)
  w.fin = w.z + (size_t)nb * nt;
  return w;
}

// One matrix group's whole sweep, enqueued on stream s.  w is already offset to the group.
static void blocked_group(ccgp_handle* h, hipStream_t s, const double* X, int n, int d, const double* y,
                          DrawView dv, int b0, int nb, int npad, double sigma2, int mean_mode,
                          double tau2, BlockedWs w, double* loglik, double* beta, int* status,
                          const BlockedPredict* pr) {
  const int nt = npad / kTile;
  {
    ScopedTimer t(h, CCGP_T_COV, s);
    launch_cov_tiles(s, X, n, d, dv, b0, nb, w.A, w.a_stride, npad, mean_mode, sigma2, tau2, w.ld);
    RhsArgs ra{w.A, w.a_stride, npad, n, y, w.ld};
    hipLaunchKernelGGL(rhs_rows_kernel, dim3(npad, nb), dim3(kTile), 0, s, ra);
    if (pr)   // rows npad+128+t = r(x_t)' (Mixed.corr.vec, HX:425-431), one row per test site
      launch_cov_cross_batched(s, pr->Xtest, pr->m, X, n, d, dv, b0, nb, w.A + npad + kTile, w.a_stride,
                               w.ld);
  }
  GemmArgs g{};
  g.A = w.A; g.a_stride = w.a_stride; g.npad = npad; g.invd = w.invd;
  g.invd_stride = (size_t)nt * kTile * kTile; g.nt = nt; g.nb = nb; g.ld = w.ld; g.ne = w.ne;
  DiagArgs dg{};
  dg.A = w.A; dg.a_stride = w.a_stride; dg.npad = npad; dg.invd = w.invd;
  dg.invd_stride = g.invd_stride; dg.logdet_part = w.z; dg.status = status + b0; dg.nt = nt;
  dg.nb = nb; dg.n = n; dg.ld = w.ld;
  static int force_s = -1;   // CCGP_STRIPS=1|2|4 pins the strip count (profiling only)
  if (force_s < 0) {
    const char* e = getenv("CCGP_STRIPS");
    force_s = e ? atoi(e) : 0;
  }
  for (int j = 0; j < nt; ++j) {
    g.j = j;
    if (j > 0) {
      ScopedTimer t(h, CCGP_T_UPDATE, s);
      g.mode = 0;
      launch_gemm(s, g, 0, force_s > 0 ? force_s : pick_strips(round_up(nb, 8) * (nt - j + 1 + w.ne)));
    }
    {
      ScopedTimer t(h, CCGP_T_DIAG, s);
      dg.j = j;
      hipLaunchKernelGGL(diag_kernel, dim3(nb), dim3(256), 0, s, dg);
    }
    {
      ScopedTimer t(h, CCGP_T_TRSM, s);
      g.mode = 1;
      launch_gemm(s, g, 1, 1);   // trsm is in place: strips of one tile would race (read-all / write-own)
    }
  }
  {
    ScopedTimer t(h, CCGP_T_SOLVE, s);
    FinishArgs fa{};
    fa.A = w.A; fa.a_stride = w.a_stride; fa.npad = npad; fa.logdet_part = w.z; fa.params = dv.params;
    fa.ldp = dv.ldp; fa.K = dv.K; fa.b0 = b0; fa.nt = nt; fa.n = n; fa.sigma2 = sigma2;
    fa.mode = mean_mode; fa.loglik = loglik; fa.beta = beta; fa.status = status; fa.ld = w.ld;
    fa.s11_out = w.fin; fa.beta_out = w.fin + nb;
    hipLaunchKernelGGL(finish_kernel, dim3(nb), dim3(256), 0, s, fa);
    if (pr) {
      PredFinishArgs pa{w.A, w.a_stride, npad, w.ld, n, pr->m, w.fin, w.fin + nb, status, b0, pr->S,
                        sigma2, pr->mean, pr->var};
      hipLaunchKernelGGL(predict_finish_kernel, dim3((pr->m + 255) / 256, nb), dim3(256), 0, s, pa);
    }
  }
}

void blocked_loglik(ccgp_handle* h, const double* X, int n, int d, const double* y, DrawView dv,
                    int b0, int nb, int npad, double sigma2, int mean_mode, double tau2,
                    BlockedWs w, double* loglik, double* beta, int* status, const BlockedPredict* pr) {
  const int nt = npad / kTile;
  static bool attr_set = false;
  if (!attr_set) {
    const void* ks[] = {(const void*)chol_update_kernel, (const void*)chol_update_s2_kernel,
                        (const void*)chol_update_s4_kernel, (const void*)chol_trsm_kernel,
                        (const void*)chol_trsm_s2_kernel, (const void*)chol_trsm_s4_kernel};
    for (const void* k : ks)
      (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes - 64);
    attr_set = true;
  }
  // split the nb matrices into independent groups (multiples of 8 matrices: the XCD-aware
  // block decode keeps 8 matrices per XCD group) and fork them onto the group streams
  int ng = h->n_groups;
  if (ng > nb / 8) ng = nb / 8;
  if (ng < 1) ng = 1;
  if (ng == 1) {
    blocked_group(h, h->stream, X, n, d, y, dv, b0, nb, npad, sigma2, mean_mode, tau2, w, loglik, beta,
                  status, pr);
    return;
  }
  (void)hipEventRecord(h->fork, h->stream);
  const int per = round_up((nb + ng - 1) / ng, 8);
  for (int gidx = 0, m0 = 0; m0 < nb; ++gidx, m0 += per) {
    const int cnt = (nb - m0) < per ? (nb - m0) : per;
    hipStream_t s = h->gstream[gidx];
    (void)hipStreamWaitEvent(s, h->fork, 0);
    BlockedWs wg = w;
    wg.A = w.A + (size_t)m0 * w.a_stride;
    wg.invd = w.invd + (size_t)m0 * nt * kTile * kTile;
    wg.z = w.z + (size_t)m0 * nt;
    wg.fin = w.fin + (size_t)2 * m0;   // (s11, beta) pairs are addressed as fin[0..cnt) and fin[cnt..2cnt)
    blocked_group(h, s, X, n, d, y, dv, b0 + m0, cnt, npad, sigma2, mean_mode, tau2, wg, loglik, beta,
                  status, pr);
    (void)hipEventRecord(h->gjoin[gidx], s);
    (void)hipStreamWaitEvent(h->stream, h->gjoin[gidx], 0);
  }
}

}  // namespace ccgp
