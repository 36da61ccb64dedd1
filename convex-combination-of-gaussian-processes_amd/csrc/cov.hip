// Covariance / convex-combination kernels.
//
// Reference arithmetic: corr.matrix HX:328-337 / ANI:351-360, corr.matrix.ISO HX:347-356,
// corr.vec(.ISO) HX:367-375 / ANI:369-377, Mixed.corr.matrix HX:408-415 / ANI:399-406,
// Mixed.corr.vec HX:425-431.  The reference's EXPANDED distance is kept:
//   dist_c(i,j) = (u_ci + u_cj) - 2 sum_k (x_ik theta_ck) x_jk,   u_ci = sum_k theta_ck x_ik^2
// (HX:352-355), so the diagonal is exp(-rounding) exactly as in R, not a forced 1.
//
// Roofline: the algorithmic traffic is the output write (8 n^2 B dense, 4 n^2 B lower tiles; X is read
// once per tile into LDS and every output column is written as 512 B contiguous per wave, lane = row),
// but the kernel is fp64-VALU bound, not HBM-bound: K (d + ~22) fp64 instructions per entry (distance,
// exp_cov, mix).  Measured at n = 4096, d = 5, K = 3: 2.2 TB/s of lower-tile writes = 0.27 of the HBM
// peak with the CUs 97 % busy (profiles/r01j).
#include "ccgp_internal.h"

namespace ccgp {

namespace {

constexpr int kCovRows = 64;   // rows per workgroup tile (one per lane)
constexpr int kCovCols = 64;   // columns per workgroup tile (16 per wave)

struct CovArgs {
  const double* A;   // m x d column-major (rows of the output)
  const double* Bm;  // n x d column-major (columns of the output)
  int m, n, d;
  const double* params;
  KernelFamily fam;
  int ldp, K;
  int draw0;
  double* out;
  size_t batch_stride;
  int ldo;
  int mode;  // 0: normalised R_mixed; 1: sigma2*sum(w^2)*R_mixed + tau2
  double sigma2, tau2;
  int lower_tiles;  // 1: square, write only tiles with row-tile >= col-tile, pad identity to npad
  int npad;
  int raw_mix;      // 1: sum w_c^2 r_c without the division by sum w_c^2 (corr.vec.combined as written, D1F:470-480)
  const double* colpad;   // SCOL instantiation: the design zero-padded to npad rows (npad x d, leading dimension npad)
  const double* upad;     // SCOL instantiation: u[z][c][i] (nb x K x npad), written by cov_u_kernel
};

// LDS: etab[256] | xa[d][64] | xb[d][64] | ua[K][64] | ub[K][64] | th[K][d] | w2[K]
// FAM = 0: Gaussian (the hot instantiation: nothing of the Matern code in it); FAM = 1: Matern (D1:348-351)
// SCOL (lower-tile launches of the blocked path): the column coordinates come from scalar loads, see below
template <int FAM, bool SCOL = false>
__global__ __launch_bounds__(256) void cov_kernel(CovArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int d = a.d, K = a.K;
  double* etab = smem;                 // 2^(j/256) for exp_cov
  double* xa = etab + kExpTableDoubles;
  double* xb = xa + d * kCovRows;
  double* ua = xb + d * kCovCols;
  double* ub = ua + K * kCovRows;
  double* th = ub + K * kCovCols;
  double* w2 = th + K * d;

  int tr = blockIdx.x, tc = blockIdx.y;
  if (a.lower_tiles) {
    // triangular launch: blockIdx.x enumerates (tr >= tc) pairs of 64-wide tiles
    int t = blockIdx.x;
    int r = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= t) ++r;
    while (r * (r + 1) / 2 > t) --r;
    tr = r;
    tc = t - r * (r + 1) / 2;
  }
  const int b = a.draw0 + blockIdx.z;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int i0 = tr * kCovRows, j0 = tc * kCovCols;

  if (FAM == 0) exp_table_load(etab, tid, 256);
  for (int e = tid; e < K * d; e += 256) th[e] = FAM == 0 ? a.params[b + (size_t)(K + e) * a.ldp] : theta_to_rate(a.fam, a.params[b + (size_t)(K + e) * a.ldp], e / d);
  if (tid < K) {
    double w = a.params[b + (size_t)tid * a.ldp];
    w2[tid] = w * w;
  }
  if constexpr (SCOL) {
    // the row sums u = X^2 Theta come from cov_u_kernel (once per draw; the update workgroups that generate their own tile
    // read the same values, so both give the same bits), the column coordinates from scalar loads: no xb at all
    for (int e = tid; e < d * kCovRows; e += 256) {
      const int k = e / kCovRows, r = e % kCovRows;
      xa[e] = a.colpad[i0 + r + (size_t)k * a.npad];
    }
    const double* ub_ = a.upad + (size_t)blockIdx.z * K * a.npad;
    for (int e = tid; e < K * kCovRows; e += 256) {
      const int c = e / kCovRows, r = e % kCovRows;
      ua[e] = ub_[(size_t)c * a.npad + i0 + r];
      ub[e] = ub_[(size_t)c * a.npad + j0 + r];
    }
    __syncthreads();
  } else {
    for (int e = tid; e < d * kCovRows; e += 256) {
      int k = e / kCovRows, r = e % kCovRows;
      int gi = i0 + r;
      xa[e] = gi < a.m ? a.A[gi + (size_t)k * a.m] : 0.0;
      int gj = j0 + r;
      xb[e] = gj < a.n ? a.Bm[gj + (size_t)k * a.n] : 0.0;
    }
    __syncthreads();
    for (int e = tid; e < K * kCovRows; e += 256) {
      int c = e / kCovRows, r = e % kCovRows;
      double sa = 0.0, sb = 0.0;
      for (int k = 0; k < d; ++k) {
        double t = th[c * d + k];
        double va = xa[k * kCovRows + r], vb = xb[k * kCovCols + r];
        sa += va * va * t;  // (X^2 %*% Theta) row sums
        sb += vb * vb * t;
      }
      ua[e] = sa;
      ub[e] = sb;
    }
    __syncthreads();
  }

  double sw = 0.0;
  for (int c = 0; c < K; ++c) sw += w2[c];
  const double post_scale = a.mode == 1 ? a.sigma2 * sw : 1.0;
  const double post_shift = a.mode == 1 ? a.tau2 : 0.0;

  double* out = a.out + (size_t)blockIdx.z * a.batch_stride;
  const int gi = i0 + lane;
  const int rows_valid = a.lower_tiles ? a.npad : a.m;
  const int cols_valid = a.lower_tiles ? a.npad : a.n;
  constexpr int JW = kCovCols / 4;   // 16 output columns per thread (one row, lane = row)
  const int jl0 = wave * JW;
  const double inv_sw = a.raw_mix ? 1.0 : 1.0 / sw;    // (sum w_c^2 R_c) / sum w_c^2 as a multiplication (<= 1 ulp apart)
  double accs[JW];
#pragma unroll
  for (int jj = 0; jj < JW; ++jj) accs[jj] = 0.0;
  // per component: the d-loop keeps the 16 dot products of this row in registers (one LDS read
  // of the row coordinate and 16 broadcast reads per dimension instead of two reads per entry).
  // Round 2 tried the dimension loop OUTSIDE a group of three components (column coordinates read once
  // per group, 3 x 16 dot products in registers: a third of the LDS reads): 179 VGPRs instead of 108, two
  // waves per SIMD instead of four, and 19.4 instead of 17.2 ms for 512 matrices -- the kernel lives on
  // occupancy (independent exp chains), not on LDS bandwidth.  Not kept.
  // Round 3: the column coordinates x_jk are WAVE-UNIFORM (a wave owns 16 columns x 64 rows).  In the lower-tile
  // launches of the blocked path they are read with scalar loads from a zero-padded copy of the design (two
  // s_load_dwordx16 per dimension into SGPRs, K$-cached; the v_fma takes them as its one constant-bus operand)
  // instead of 16 broadcast ds_reads per dimension and component: with three components the LDS pipe was as busy
  // as the VALU (21 ds_read_b64 per entry against 57 fp64 instructions).
  typedef double d8 __attribute__((ext_vector_type(8)));
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const double* __restrict__ bcol = SCOL ? a.colpad + j0 + wave_u * JW : nullptr;
  for (int c = 0; c < K; ++c) {
    double sdot[JW];
#pragma unroll
    for (int jj = 0; jj < JW; ++jj) sdot[jj] = 0.0;
    for (int k = 0; k < d; ++k) {
      const double xr = xa[k * kCovRows + lane] * th[c * d + k];
      if constexpr (SCOL) {
        static_assert(JW == 16, "two 8-double scalar loads per dimension");
        const d8 b0 = *reinterpret_cast<const d8*>(bcol + (size_t)k * a.npad);
        const d8 b1 = *reinterpret_cast<const d8*>(bcol + (size_t)k * a.npad + 8);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          sdot[jj] = fma(xr, b0[jj], sdot[jj]);
          sdot[8 + jj] = fma(xr, b1[jj], sdot[8 + jj]);
        }
      } else {
#pragma unroll
        for (int jj = 0; jj < JW; ++jj) sdot[jj] = fma(xr, xb[k * kCovCols + jl0 + jj], sdot[jj]);
      }
    }
    const double ur = ua[c * kCovRows + lane], wc = w2[c];
#pragma unroll
    for (int jj = 0; jj < JW; ++jj) {
      if constexpr (FAM == 0) {
        accs[jj] = cov_mix_term(accs[jj], wc, ur, ub[c * kCovCols + jl0 + jj], sdot[jj], etab);
      } else {
        const double dist = (ur + ub[c * kCovCols + jl0 + jj]) + (-2.0 * sdot[jj]);
        accs[jj] = fma(wc, corr_of_dist(a.fam, dist, etab, c), accs[jj]);
      }
    }
  }
  if constexpr (SCOL) {
    // lower-tile launches (the blocked path's matrix build): the tile lies inside the npad x npad array (npad a multiple
    // of 64), so no bounds tests; the identity padding exists only in the tiles that reach past n -- a wave-uniform
    // branch; and a column's address is base + lane * 8 + (column * ld * 8 in an SGPR): one buffer_store per output and
    // NO vector address arithmetic.  Before, every output paid a 64-bit multiply-add for its address, two compares and
    // selects for the padding and an exec-mask branch for the bounds: ~20 of the 102 VALU instructions per entry of a
    // kernel whose VALU is busy 96 % of the time (profiles/r04/pmc_cov_summary.json).
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, -1, 0x00020000);
    const unsigned voff = (unsigned)gi * 8u;
    const unsigned col0 = (unsigned)(j0 + wave_u * JW);
    const double scale = post_scale * inv_sw;
    auto store = [&](int jj, double v) {
      u2v bits;
      bits[0] = (unsigned)__double2loint(v);
      bits[1] = (unsigned)__double2hiint(v);
      __builtin_amdgcn_raw_buffer_store_b64(bits, rout, voff, (col0 + (unsigned)jj) * (unsigned)a.ldo * 8u, 0);
    };
    if (i0 + kCovRows > a.n || j0 + kCovCols > a.n) {   // block-uniform: the tiles that reach into the identity padding
#pragma unroll
      for (int jj = 0; jj < JW; ++jj) {
        const int gj = (int)col0 + jj;
        double v = fma(scale, accs[jj], post_shift);
        if (gi >= a.n || gj >= a.n) v = (gi == gj) ? 1.0 : 0.0;
        store(jj, v);
      }
    } else {
#pragma unroll
      for (int jj = 0; jj < JW; ++jj) store(jj, fma(scale, accs[jj], post_shift));
    }
  } else {
#pragma unroll
    for (int jj = 0; jj < JW; ++jj) {
      const int gj = j0 + jl0 + jj;
      if (gj >= cols_valid) break;
      double v = fma(post_scale * inv_sw, accs[jj], post_shift);
      if (a.lower_tiles) {
        if (gi >= a.n || gj >= a.n) v = (gi == gj) ? 1.0 : 0.0;  // identity padding
      }
      if (gi < rows_valid) out[gi + (size_t)gj * a.ldo] = v;
    }
  }
}

size_t cov_lds(int d, int K) {
  return sizeof(double) * (size_t)(kExpTableDoubles + 2 * d * 64 + 2 * K * 64 + K * d + K);
}

// d = 64, K = 8 needs 78 KiB: above the 64 KiB a kernel gets without asking
void cov_prepare() {
  static unsigned long long attr_mask = 0;
  once_per_device(attr_mask, [] {
    raise_lds_limit((const void*)cov_kernel<0>, "cov_kernel<0>");
    raise_lds_limit((const void*)cov_kernel<0, true>, "cov_kernel<0, scalar columns>");
    raise_lds_limit((const void*)cov_kernel<1>, "cov_kernel<1>");
  });
}

}  // namespace

void launch_cov_dense(hipStream_t s, const double* A, int m, const double* Bm, int n, int d,
                      DrawView dv, int draw, double* out, int ldo) {
  CovArgs a{};
  a.A = A; a.Bm = Bm; a.m = m; a.n = n; a.d = d;
  a.params = dv.params; a.ldp = dv.ldp; a.K = dv.K; a.fam = dv.fam; a.draw0 = draw;
  a.out = out; a.batch_stride = 0; a.ldo = ldo; a.mode = 0; a.lower_tiles = 0; a.npad = 0;
  a.raw_mix = dv.fam.id == 2 && A != Bm;   // the two-family script's corr.vec.combined never divides (D1F:479)
  cov_prepare();
  dim3 grid((m + kCovRows - 1) / kCovRows, (n + kCovCols - 1) / kCovCols, 1);
  if (dv.fam.id == 0) hipLaunchKernelGGL(cov_kernel<0>, grid, dim3(256), cov_lds(d, dv.K), s, a);
  else hipLaunchKernelGGL(cov_kernel<1>, grid, dim3(256), cov_lds(d, dv.K), s, a);
}

namespace {
// the design zero-padded to npad rows: xpad[i + k * npad]
__global__ void pad_design_kernel(const double* X, int n, int d, int npad, double* xpad) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= npad * d) return;
  const int k = e / npad, i = e % npad;
  xpad[e] = i < n ? X[i + (size_t)k * n] : 0.0;
}
}  // namespace

namespace {
// u[z][c][i] = sum_k theta_ck x_ik^2 (the row sums of X^2 Theta, HX:352-353) for draw b0 + z: zero on the padding rows
__global__ void cov_u_kernel(const double* xpad, int npad, int d, const double* params, int ldp, int K, int b0, double* upad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, z = blockIdx.z;
  if (i >= npad) return;
  double s = 0.0;
  for (int k = 0; k < d; ++k) {
    const double v = xpad[i + (size_t)k * npad];
    s = fma(v * v, params[b0 + z + (size_t)(K + c * d + k) * ldp], s);
  }
  upad[((size_t)z * K + c) * npad + i] = s;
}
}  // namespace

void launch_cov_tiles(hipStream_t s, const double* X, int n, int d, DrawView dv, int b0, int nb,
                      double* Abase, size_t batch_stride, int npad, int mean_mode, double sigma2,
                      double tau2, int ld, double* xpad, double* upad) {
  CovArgs a{};
  a.A = X; a.Bm = X; a.m = n; a.n = n; a.d = d;
  a.params = dv.params; a.ldp = dv.ldp; a.K = dv.K; a.fam = dv.fam; a.draw0 = b0;
  a.out = Abase; a.batch_stride = batch_stride; a.ldo = ld; a.mode = mean_mode;
  a.sigma2 = sigma2; a.tau2 = tau2; a.lower_tiles = 1; a.npad = npad;
  cov_prepare();
  int nt64 = npad / 64;
  dim3 grid(nt64 * (nt64 + 1) / 2, 1, nb);
  // the scalar-column instantiation addresses a matrix through 32-bit buffer offsets
  const bool fits32 = (size_t)npad * (size_t)ld * 8 < 0xFFFF0000ull;
  if (dv.fam.id == 0 && xpad && upad && fits32) {
    hipLaunchKernelGGL(pad_design_kernel, dim3((npad * d + 255) / 256), dim3(256), 0, s, X, n, d, npad, xpad);
    hipLaunchKernelGGL(cov_u_kernel, dim3((npad + 255) / 256, dv.K, nb), dim3(256), 0, s, xpad, npad, d, dv.params, dv.ldp,
                       dv.K, b0, upad);
    a.colpad = xpad;
    a.upad = upad;
    if (grid.x) hipLaunchKernelGGL((cov_kernel<0, true>), grid, dim3(256), cov_lds(d, dv.K), s, a);
  } else if (dv.fam.id == 0) {
    hipLaunchKernelGGL(cov_kernel<0>, grid, dim3(256), cov_lds(d, dv.K), s, a);
  } else {
    hipLaunchKernelGGL(cov_kernel<1>, grid, dim3(256), cov_lds(d, dv.K), s, a);
  }
}

// Batched cross-correlation rows for the blocked prediction path: for draw b0+z the m x n block
// out[t + i*ldo] = R_mixed(x_t, x_i) (normalised, no scale/shift) at Abase + z*batch_stride.
void launch_cov_cross_batched(hipStream_t s, const double* Xtest, int m, const double* X, int n, int d,
                              DrawView dv, int b0, int nb, double* Abase, size_t batch_stride, int ldo) {
  CovArgs a{};
  a.A = Xtest; a.Bm = X; a.m = m; a.n = n; a.d = d;
  a.params = dv.params; a.ldp = dv.ldp; a.K = dv.K; a.fam = dv.fam; a.draw0 = b0;
  a.out = Abase; a.batch_stride = batch_stride; a.ldo = ldo; a.mode = 0; a.lower_tiles = 0; a.npad = 0;
  a.raw_mix = dv.fam.id == 2;
  cov_prepare();
  dim3 grid((m + kCovRows - 1) / kCovRows, (n + kCovCols - 1) / kCovCols, nb);
  if (dv.fam.id == 0) hipLaunchKernelGGL(cov_kernel<0>, grid, dim3(256), cov_lds(d, dv.K), s, a);
  else hipLaunchKernelGGL(cov_kernel<1>, grid, dim3(256), cov_lds(d, dv.K), s, a);
}

}  // namespace ccgp
