// Batched blocked Cholesky + solves for n > 128 (BASELINE config 4: n = 4096).
//
// LEFT-looking, 128-wide block columns, many matrices per launch.  For block column j:
//   update : T_ij = A_ij - sum_{k<j} L_ik L_jk'      all tiles i >= j   (f64 MFMA GEMM)
//   diag   : L_jj = chol(T_jj), W_j = L_jj^-1        one workgroup per matrix, in LDS
//   trsm   : L_ij = T_ij W_j'                        tiles i > j        (f64 MFMA GEMM)
// then a forward substitution for [y 1] and the reductions that the reference's
// dmnorm / beta.MLE need (HX:458-460, HX:570).  Left-looking because every tile is then
// written once and the panels it re-reads are shared through L2 / Infinity Cache: HBM
// traffic is ~2 x 4n^2 B per matrix instead of the right-looking 8n^3/(3 nb) B, so the
// trailing update is MFMA-bound, not HBM-bound.
//
// MFMA: v_mfma_f64_16x16x4_f64.  Operand lane map (one f64 per lane):
//   A[i = lane&15][k = lane>>4],  B[k = lane>>4][j = lane&15],
//   D[i = (lane>>4) + 4*r][j = lane&15] for accumulator register r = 0..3.
// We feed A := Q (the block-column operand) and B := P (the block-row operand), so D's
// lane index runs along matrix ROWS: a wave's store is 4 columns x 128 contiguous bytes.
#include "ccgp_internal.h"

namespace ccgp {

typedef double d4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kBK = 16;              // k-depth of one LDS stage
constexpr int kLdsRow = kTile + 16;  // padded row (doubles): 1152 B, conflict-free ds_read_b64
constexpr int kStageDoubles = 2 * kBK * kLdsRow;

struct GemmArgs {
  double* A;
  size_t a_stride;
  int npad;
  double* invd;
  size_t invd_stride;
  int j, nt, nb;
  int mode;  // 0: update, 1: trsm
};

__device__ inline void decode_block(int L, int ntile, int nb, int& b, int& t) {
  // XCD-aware: blocks are dealt round-robin over the 8 XCDs, so L % 8 labels the XCD
  // group.  Keep every tile of one matrix in one group: they share the Q panel in L2.
  int grp = L / (8 * ntile), r = L % (8 * ntile);
  b = grp * 8 + (r & 7);
  t = r >> 3;
  (void)nb;
}

__global__ __launch_bounds__(256, 2) void tile_gemm_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int ntile = g.mode == 0 ? g.nt - g.j : g.nt - g.j - 1;
  int b, t;
  decode_block(blockIdx.x, ntile, g.nb, b, t);
  if (b >= g.nb) return;
  const int i = g.j + t + (g.mode == 0 ? 0 : 1);
  double* Ab = g.A + (size_t)b * g.a_stride;
  const int ld = g.npad;

  const double* P;
  const double* Q;
  int ldP, ldQ, Kdim;
  if (g.mode == 0) {
    P = Ab + (size_t)i * kTile;
    Q = Ab + (size_t)g.j * kTile;
    ldP = ldQ = ld;
    Kdim = g.j * kTile;
  } else {
    P = Ab + (size_t)i * kTile + (size_t)g.j * kTile * ld;
    Q = g.invd + (size_t)b * g.invd_stride + (size_t)g.j * kTile * kTile;
    ldP = ld;
    ldQ = kTile;
    Kdim = kTile;
  }
  double* C = Ab + (size_t)i * kTile + (size_t)g.j * kTile * ld;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, l4 = lane >> 4;

  d4 acc[4][4];  // [n-subtile][m-subtile]
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = d4{0.0, 0.0, 0.0, 0.0};

  // staging: thread loads 4 x double2 of P and of Q per stage; element e = tid + 256 q
  // -> k = e / 64 (= wave + 4 q), rows 2*(e%64), +1  (one wave = one 1 KiB column)
  const int r2 = (tid & 63) * 2;
  double2 pr[4], qr[4];
  auto gload = [&](int kt) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = kt * kBK + wave + 4 * q;
      pr[q] = *reinterpret_cast<const double2*>(P + r2 + (size_t)k * ldP);
      qr[q] = *reinterpret_cast<const double2*>(Q + r2 + (size_t)k * ldQ);
    }
  };
  auto lstore = [&](int stage) {
    double* Ps = smem + stage * kStageDoubles;
    double* Qs = Ps + kBK * kLdsRow;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = wave + 4 * q;
      *reinterpret_cast<double2*>(Ps + k * kLdsRow + r2) = pr[q];
      *reinterpret_cast<double2*>(Qs + k * kLdsRow + r2) = qr[q];
    }
  };

  const int nk = Kdim / kBK;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload(kt + 1);
    const double* Ps = smem + (kt & 1) * kStageDoubles;
    const double* Qs = Ps + kBK * kLdsRow;
#pragma unroll
    for (int kk = 0; kk < kBK / 4; ++kk) {
      double pf[4], qf[4];
      const int krow = (kk * 4 + l4) * kLdsRow;
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        pf[x] = Ps[krow + wm * 64 + x * 16 + l15];
        qf[x] = Qs[krow + wn * 64 + x * 16 + l15];
      }
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
          acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(qf[x], pf[y], acc[x][y], 0, 0, 0);
    }
    if (kt + 1 < nk) lstore((kt + 1) & 1);
    __syncthreads();
  }

  // epilogue: D[(l4 + 4r)][l15] of sub-tile (x = column block, y = row block)
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = wn * 64 + x * 16 + l4 + 4 * r;
        const int row = wm * 64 + y * 16 + l15;
        double* p = C + row + (size_t)col * ld;
        if (g.mode == 0) *p = *p - acc[x][y][r];
        else *p = acc[x][y][r];
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
};

__global__ __launch_bounds__(256) void diag_kernel(DiagArgs g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* T = smem;                 // 128 x 128 column-major
  double* tmp = T + kTile * kTile;  // 128
  double* red = tmp + kTile;        // 8
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* Ab = g.A + (size_t)b * g.a_stride;
  double* C = Ab + (size_t)g.j * kTile + (size_t)g.j * kTile * g.npad;
  const double kNaN = __longlong_as_double(0x7ff8000000000000LL);

  for (int e = tid; e < kTile * kTile; e += 256) {
    int r = e & (kTile - 1), c = e >> 7;
    T[e] = r >= c ? C[r + (size_t)c * g.npad] : 0.0;
  }
  int bad = 0;
  for (int k = 0; k < kTile; ++k) {
    __syncthreads();
    const double piv = T[k + k * kTile];
    if (!(piv > 0.0)) { bad = k + 1; break; }
    const double rinv = 1.0 / piv;
    const double* colk = T + k * kTile;
    for (int c = k + 1 + wave; c < kTile; c += 4) {
      const double lck = colk[c] * rinv;
      double* colc = T + c * kTile;
      for (int r = c + lane; r < kTile; r += 64) colc[r] = fma(-colk[r], lck, colc[r]);
    }
  }
  __syncthreads();
  // scale columns: L[r][k] = T[r][k] / sqrt(d_k); accumulate log d_k
  double lsum = 0.0;
  for (int k = wave; k < kTile; k += 4) {
    const double dk = T[k + k * kTile];
    const double rs = 1.0 / sqrt(dk);
    if (lane == 0) lsum += log(dk);
    for (int r = k + lane; r < kTile; r += 64) {
      double v = T[r + k * kTile];
      T[r + k * kTile] = (r == k) ? sqrt(dk) : v * rs;
    }
  }
  if (lane == 0) red[wave] = lsum;
  __syncthreads();
  if (tid == 0) {
    g.logdet_part[(size_t)b * g.nt + g.j] = bad ? kNaN : (red[0] + red[1] + red[2] + red[3]);
    if (bad && g.status[b] == 0) g.status[b] = g.j * kTile + bad;
  }
  // write L (poisoned with NaN on failure so that the likelihood comes out NaN)
  for (int e = tid; e < kTile * kTile; e += 256) {
    int r = e & (kTile - 1), c = e >> 7;
    C[r + (size_t)c * g.npad] = bad ? kNaN : T[e];
  }
  __syncthreads();
  // in-place inverse of the lower-triangular L (column sweep from the right, dtrti2 order):
  //   x_jj = 1 / l_jj ;  x[j+1:, j] = -x_jj * X[j+1:, j+1:] * l[j+1:, j]
  for (int j = kTile - 1; j >= 0; --j) {
    if (tid < kTile) tmp[tid] = T[tid + j * kTile];
    __syncthreads();
    const double xjj = 1.0 / tmp[j];
    if (tid < kTile) {
      const int r = tid;
      if (r == j) T[r + j * kTile] = xjj;
      else if (r > j) {
        double s = 0.0;
        for (int k = j + 1; k <= r; ++k) s = fma(T[r + k * kTile], tmp[k], s);
        T[r + j * kTile] = -xjj * s;
      }
    }
    __syncthreads();
  }
  double* W = g.invd + (size_t)b * g.invd_stride + (size_t)g.j * kTile * kTile;
  for (int e = tid; e < kTile * kTile; e += 256) W[e] = bad ? kNaN : T[e];
}

// ---- forward substitution for [y 1] and the likelihood reductions --------------------------
struct SolveArgs {
  const double* A;
  size_t a_stride;
  int npad;
  const double* invd;
  size_t invd_stride;
  const double* logdet_part;
  const double* y;
  const double* params;
  int ldp, K;
  int b0, nt, nb, n;
  double sigma2;
  int mode;
  double* loglik;
  double* beta;
  const int* status;
};

__device__ inline double block_sum(double v, double* red, int tid) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void solve_kernel(SolveArgs g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int npad = g.npad;
  double* zy = smem;              // npad
  double* z1 = zy + npad;         // npad
  double* part = z1 + npad;       // 2 x 2 x 128 partial sums
  double* rhs = part + 4 * kTile; // 2 x 128
  double* red = rhs + 2 * kTile;  // 8
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int r = tid & (kTile - 1), half = tid >> 7;
  const double* Ab = g.A + (size_t)b * g.a_stride;
  const double* Wb = g.invd + (size_t)b * g.invd_stride;

  for (int jb = 0; jb < g.nt; ++jb) {
    // rhs = b_jb - sum_{k < jb*128} L[jb*128 + r, k] z[k]   (k split in halves by parity)
    double sy = 0.0, s1 = 0.0;
    const double* Lrow = Ab + (size_t)jb * kTile + r;
    const int kend = jb * kTile;
#pragma unroll 8
    for (int k = half; k < kend; k += 2) {
      const double l = Lrow[(size_t)k * npad];
      sy = fma(l, zy[k], sy);
      s1 = fma(l, z1[k], s1);
    }
    part[half * 2 * kTile + r] = sy;
    part[half * 2 * kTile + kTile + r] = s1;
    __syncthreads();
    if (tid < kTile) {
      const int gi = jb * kTile + r;
      rhs[r] = (gi < g.n ? g.y[gi] : 0.0) - (part[r] + part[2 * kTile + r]);
      rhs[kTile + r] = (gi < g.n ? 1.0 : 0.0) - (part[kTile + r] + part[3 * kTile + r]);
    }
    __syncthreads();
    // z_jb = W_jb rhs   (W lower triangular, column-major, ld 128); halves split k by parity
    const double* W = Wb + (size_t)jb * kTile * kTile;
    sy = 0.0; s1 = 0.0;
#pragma unroll 8
    for (int k = half; k <= r; k += 2) {
      const double w = W[r + k * kTile];
      sy = fma(w, rhs[k], sy);
      s1 = fma(w, rhs[kTile + k], s1);
    }
    __syncthreads();
    part[half * 2 * kTile + r] = sy;
    part[half * 2 * kTile + kTile + r] = s1;
    __syncthreads();
    if (tid < kTile) {
      zy[jb * kTile + r] = part[r] + part[2 * kTile + r];
      z1[jb * kTile + r] = part[kTile + r] + part[3 * kTile + r];
    }
    __syncthreads();
  }

  double logdet = 0.0;
  for (int jb = tid; jb < g.nt; jb += 256) logdet += g.logdet_part[(size_t)b * g.nt + jb];
  logdet = block_sum(logdet, red, tid);
  double s11 = 0.0, s1y = 0.0, syy = 0.0;
  for (int k = tid; k < g.n; k += 256) {
    s11 = fma(z1[k], z1[k], s11);
    s1y = fma(z1[k], zy[k], s1y);
    syy = fma(zy[k], zy[k], syy);
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
    for (int k = tid; k < g.n; k += 256) { double v = zy[k] - beta * z1[k]; q = fma(v, v, q); }
    q = block_sum(q, red, tid);
    ll = -0.5 * (g.n * kLog2Pi + g.n * log(cs) + logdet + q / cs);
  } else {
    ll = -0.5 * (g.n * kLog2Pi + logdet + syy);
  }
  if (tid == 0) {
    const double kNaN = __longlong_as_double(0x7ff8000000000000LL);
    if (g.status && g.status[gb] != 0) { ll = kNaN; beta = kNaN; }
    g.loglik[gb] = ll;
    if (g.beta) g.beta[gb] = beta;
  }
}

}  // namespace

size_t blocked_ws_bytes(int npad, int nb) {
  const int nt = npad / kTile;
  size_t dbl = (size_t)nb * npad * npad + (size_t)nb * nt * kTile * kTile + (size_t)nb * nt + 64;
  return dbl * sizeof(double);
}

BlockedWs blocked_carve(void* ws, int npad, int nb) {
  const int nt = npad / kTile;
  BlockedWs w{};
  w.A = static_cast<double*>(ws);
  w.a_stride = (size_t)npad * npad;
  w.invd = w.A + (size_t)nb * w.a_stride;
  w.z = w.invd + (size_t)nb * nt * kTile * kTile;  // logdet partials live here (nb x nt)
  return w;
}

void blocked_loglik(ccgp_handle* h, const double* X, int n, int d, const double* y, DrawView dv,
                    int b0, int nb, int npad, double sigma2, int mean_mode, double tau2,
                    BlockedWs w, double* loglik, double* beta, int* status) {
  hipStream_t s = h->stream;
  const int nt = npad / kTile;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)tile_gemm_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes - 64);
    (void)hipFuncSetAttribute((const void*)diag_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes - 64);
    (void)hipFuncSetAttribute((const void*)solve_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes - 64);
    attr_set = true;
  }
  {
    ScopedTimer t(h, CCGP_T_COV);
    launch_cov_tiles(s, X, n, d, dv, b0, nb, w.A, w.a_stride, npad, mean_mode, sigma2, tau2);
  }
  GemmArgs g{};
  g.A = w.A; g.a_stride = w.a_stride; g.npad = npad; g.invd = w.invd;
  g.invd_stride = (size_t)nt * kTile * kTile; g.nt = nt; g.nb = nb;
  DiagArgs dg{};
  dg.A = w.A; dg.a_stride = w.a_stride; dg.npad = npad; dg.invd = w.invd;
  dg.invd_stride = g.invd_stride; dg.logdet_part = w.z; dg.status = status + b0; dg.nt = nt;
  dg.nb = nb; dg.n = n;
  const size_t gemm_lds = 2 * kStageDoubles * sizeof(double);
  const size_t diag_lds = (kTile * kTile + kTile + 8) * sizeof(double);
  const int nb8 = round_up(nb, 8);
  for (int j = 0; j < nt; ++j) {
    g.j = j;
    if (j > 0) {
      ScopedTimer t(h, CCGP_T_UPDATE);
      g.mode = 0;
      hipLaunchKernelGGL(tile_gemm_kernel, dim3(nb8 * (nt - j)), dim3(256), gemm_lds, s, g);
    }
    {
      ScopedTimer t(h, CCGP_T_DIAG);
      dg.j = j;
      hipLaunchKernelGGL(diag_kernel, dim3(nb), dim3(256), diag_lds, s, dg);
    }
    if (j + 1 < nt) {
      ScopedTimer t(h, CCGP_T_TRSM);
      g.mode = 1;
      hipLaunchKernelGGL(tile_gemm_kernel, dim3(nb8 * (nt - j - 1)), dim3(256), gemm_lds, s, g);
    }
  }
  {
    ScopedTimer t(h, CCGP_T_SOLVE);
    SolveArgs sa{};
    sa.A = w.A; sa.a_stride = w.a_stride; sa.npad = npad; sa.invd = w.invd;
    sa.invd_stride = g.invd_stride; sa.logdet_part = w.z; sa.y = y; sa.params = dv.params;
    sa.ldp = dv.ldp; sa.K = dv.K; sa.b0 = b0; sa.nt = nt; sa.nb = nb; sa.n = n;
    sa.sigma2 = sigma2; sa.mode = mean_mode; sa.loglik = loglik; sa.beta = beta; sa.status = status;
    const size_t solve_lds = ((size_t)2 * npad + 6 * kTile + 8) * sizeof(double);
    hipLaunchKernelGGL(solve_kernel, dim3(nb), dim3(256), solve_lds, s, sa);
  }
}

}  // namespace ccgp
