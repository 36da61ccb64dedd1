// EXPERIMENT (round 1g, not part of libccgp): MFMA-blocked replacement for diag_kernel in
// csrc/blocked.hip (128 x 128 diagonal block: Cholesky factor + inverse).  It passed every blocked-path
// parity test when wired in, and was measured at 65 us per launch against 68 us for the register sweep
// it would replace (profiles/r01g_update_kernel_experiments.md, section 6) -- not enough to carry a
// second 250-line kernel, so it is parked here as a starting point: the limit is the 16-step dependent
// chain of the 16 x 16 sweep (S1, ~550 cycles per step on one wave).  To try it again: paste this block
// in front of "right-hand-side rows" in blocked.hip, launch it with d16::lds_bytes() of dynamic LDS in
// place of diag_kernel, and add it to the hipFuncSetAttribute list.
// ---- diagonal block, MFMA-blocked variant --------------------------------------------------------
// The register sweep above pays 128 dependent column steps of ~1200 cycles each (65 us).  Here the
// block is factorised right-looking in 16 x 16 sub-blocks held in (swizzled) LDS:
//   S1  one wave factorises the 16 x 16 diagonal sub-block AND inverts it in the same sweep (identity
//       appended as extra rows): 16 lanes hold one row each, the pivot row travels through
//       v_readlane -> SGPR operands, no LDS round trip and no barrier inside the 16 steps;
//   S2  panel:    L[i][k] = M[i][k] V',  E[t][k] = E[t][k] V'          (V = L_kk^-1, MFMA 16x16x4)
//   S3  trailing: M[i][j] -= L[i][k] L[j][k]',  E[t][j] -= E[t][k] L[j][k]'   (MFMA 16x16x4)
// E starts as the identity appended below the block, so after the sweep E = L^-T: the inverse costs
// the same MFMA updates as the factorisation instead of a second substitution pass.  E is upper
// triangular and lives in the UPPER sub-blocks of the same LDS array (M only needs the lower ones);
// its eight diagonal sub-blocks go to a side buffer.  Every product is of the form P Q' with both
// operands read along rows, i.e. as conflict-free 16-lane fragments.
namespace d16 {
constexpr int kMs = kTile * kTile;       // swizzled 128 x 128
constexpr int kEd = 8 * 256;             // E's diagonal sub-blocks, element (t, c) of block b at b*256 + c*16 + t
constexpr int kVb = 256;                 // V as Q operand: element (c, k) at k*16 + c
constexpr size_t lds_bytes() { return sizeof(double) * (kMs + kEd + kVb + kTile) + 64; }
__device__ __forceinline__ int idx(int r, int c) { return c * kTile + ((((r >> 4) ^ (c & 1)) << 4) | (r & 15)); }
// value of lane L of every 16-lane row, to all lanes of that row (v_mov_b32_dpp row_newbcast:L): the
// pivot row of the 16 x 16 sweep travels through VGPRs; v_readlane -> SGPR operands ran out of SGPRs
// and were spilled back into VGPR lanes (94 of the kernel's 187 k cycles)
template <int L>
__device__ __forceinline__ double row_bcast(double v) {
  return __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), 0x150 + L, 0xf, 0xf, false),
                          __builtin_amdgcn_mov_dpp(__double2loint(v), 0x150 + L, 0xf, 0xf, false));
}
// columns C = K+1 .. 15 of step K of the (square-root free) sweep:
//   D[r][C] -= (D[r][K] / p) D[C][K],   X[r][C] -= (X[r][K] / p) D[C][K]
// The broadcast D[C][K] does not wait for 1/p, so the DPP moves run underneath the reciprocal chain.
template <int K, int C>
__device__ __forceinline__ void s1_cols(double (&Dr)[16], double (&Xr)[16], double colk, double ls, double xs) {
  if constexpr (C < 16) {
    const double dc = row_bcast<C>(colk);
    Dr[C] = fma(-ls, dc, Dr[C]);
    Xr[C] = fma(-xs, dc, Xr[C]);
    s1_cols<K, C + 1>(Dr, Xr, colk, ls, xs);
  }
}
// steps K .. 15 of the 16 x 16 factorisation A = L' D L'^T with the identity appended as extra rows:
// lane r (of every 16-lane row) holds row r of the block in Dr and row r of the appended identity in
// Xr.  Afterwards Dr[K] = L'[r][K] p_K (r > K), Xr = row r of L'^-T, P[K] = p_K.  Returns 0 or
// 1 + the first non-positive pivot.
template <int K>
__device__ __forceinline__ int s1_steps(double (&Dr)[16], double (&Xr)[16], double (&P)[16], int r, int badk) {
  if constexpr (K < 16) {
    const double p = row_bcast<K>(Dr[K]);
    if (!(p > 0.0) && badk == 0) badk = K + 1;   // no early exit (straight-line code); the NaNs that follow are discarded
    P[K] = p;
    const double colk = r > K ? Dr[K] : 0.0;     // rows <= K are finished
    double rinv = __builtin_amdgcn_rcp(p);
    rinv = fma(fma(-p, rinv, 1.0), rinv, rinv);
    rinv = fma(fma(-p, rinv, 1.0), rinv, rinv);
    s1_cols<K, K + 1>(Dr, Xr, colk, colk * rinv, Xr[K] * rinv);
    return s1_steps<K + 1>(Dr, Xr, P, r, badk);
  } else {
    return badk;
  }
}
}  // namespace d16

__global__ __launch_bounds__(256, 1) void diag16_kernel(DiagArgs g) {
  using namespace d16;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Ms = smem;
  double* Ed = Ms + kMs;
  double* Vb = Ed + kEd;
  double* dvec = Vb + kVb;
  int* flag = reinterpret_cast<int*>(dvec + kTile);
  __shared__ double red[4];
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int ld = g.ld;
  double* C = g.A + (size_t)b * g.a_stride + (size_t)g.j * kTile + (size_t)g.j * kTile * ld;
  const double kNaN = __longlong_as_double(0x7ff8000000000000LL);

  // ---- load: lower sub-blocks (whole diagonal sub-blocks) of T_jj; zero where E will grow
  {
    // all 64 loads of a thread in flight before the first LDS store (one memory latency, not 16)
    const int r = tid & 127, ch = tid >> 7;
    double v[kTile / 2];
#pragma unroll
    for (int i = 0; i < kTile / 2; ++i) {
      const int c = 2 * i + ch;
      v[i] = (r >> 4) >= (c >> 4) ? C[r + (size_t)c * ld] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < kTile / 2; ++i) Ms[idx(r, 2 * i + ch)] = v[i];
    if (tid == 0) flag[0] = 0;
  }
  __syncthreads();

  // 16 x 16 sub-block (R, Cb) of the main array.  Element (row l15, column 4q + l4) -- the MFMA row
  // fragment of k-step q AND accumulator register q of the result layout -- sits at blk(R, Cb) + 512 q:
  // the column parity is l4 & 1 for every q, so the swizzle is one XOR of a per-lane constant with R << 4.
  const int lane_off = l4 * kTile + l15, sw16 = (l4 & 1) << 4;
  auto blk = [&](int R, int Cb) { return Ms + (Cb * 2048 + lane_off + ((R << 4) ^ sw16)); };
  const double* ed_frag = Ed + l4 * 16 + l15;   // + kb * 256 + 64 q
  const double* vb_frag = Vb + l4 * 16 + l15;   // + 64 q

  // S1: wave 0 factorises diagonal sub-block kb (rows of the block / of the appended identity per lane)
  auto s1 = [&](int kb) {
    double Dr[16], Xr[16], P[16];
    const int r = l15;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      Dr[c] = Ms[idx(16 * kb + r, 16 * kb + c)];
      Xr[c] = c == r ? 1.0 : 0.0;
    }
    const int badk = s1_steps<0>(Dr, Xr, P, r, 0);   // the four 16-lane rows run the same sweep
    if (badk) {
      if (lane == 0) flag[0] = 16 * kb + badk;
    } else if (lane < 16) {
      // L = L' D^1/2 back (zeros above the diagonal), E's diagonal sub-block L_kk^-T = L'^-T D^-1/2, and
      // V = L_kk^-1 as Q operand.  1/sqrt(p_c): v_rsq_f64 + two Newton steps; sqrt(p_c) = p_c y + Heron.
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const double pc = P[c];
        double y = __builtin_amdgcn_rsq(pc);
        y = fma(0.5 * y, fma(-(pc * y), y, 1.0), y);
        y = fma(0.5 * y, fma(-(pc * y), y, 1.0), y);
        double sq = pc * y;
        sq = fma(0.5 * y, fma(-sq, sq, pc), sq);
        if (r == c) dvec[16 * kb + c] = pc;
        Ms[idx(16 * kb + r, 16 * kb + c)] = c < r ? Dr[c] * y : (c == r ? sq : 0.0);
        const double x = c >= r ? Xr[c] * y : 0.0;            // (L_kk^-T)[r][c]
        Ed[kb * 256 + c * 16 + r] = x;                        // element (t = r, c)
        Vb[r * 16 + c] = x;                                   // V[c][k = r] = X[r][c]: element (c, k) at k*16 + c
      }
    }
  };
  // S3 for one or two sub-blocks:  block(s, jb) -= P_s Q_jb',  P_s = block (s, kb) (E's own diagonal
  // sub-block from the side buffer), Q_jb = L[jb][kb].  Two at a time so that their LDS round trips
  // and their dependent MFMA chains overlap.
  auto s3 = [&](int kb, int s0, int j0, int s1_, int j1, bool two) {
    double pf[2][4], qf[2][4], cv[2][4];
    const int ss[2] = {s0, s1_}, jj[2] = {j0, j1};
    double* cb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (q == 1 && !two) break;
      const double* pp = ss[q] == kb ? ed_frag + kb * 256 : blk(ss[q], kb);
      const int pstep = ss[q] == kb ? 64 : 512;
      const double* qp = blk(jj[q], kb);
      cb[q] = blk(ss[q], jj[q]);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        pf[q][kk] = pp[pstep * kk];
        qf[q][kk] = qp[512 * kk];
        cv[q][kk] = cb[q][512 * kk];
      }
    }
    d4 acc[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(qf[0][kk], pf[0][kk], acc[0], 0, 0, 0);
      if (two) acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(qf[1][kk], pf[1][kk], acc[1], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (q == 1 && !two) break;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) cb[q][512 * rr] = cv[q][rr] - acc[q][rr];
    }
  };

  int bad = 0;
  if (wave == 0) s1(0);
  __syncthreads();
  for (int kb = 0; kb < 8; ++kb) {
    bad = flag[0];
    if (bad) break;   // uniform (read behind a barrier)

    // ---- S2: panel.  7 sub-blocks: rows kb+1..7 of M, rows 0..kb-1 of E (its row kb is Ed[kb] already)
    for (int u = wave; u < 7; u += 4) {
      const int R = u < 7 - kb ? kb + 1 + u : u - (7 - kb);
      double* pb = blk(R, kb);
      d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(vb_frag[64 * kk], pb[512 * kk], acc, 0, 0, 0);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) pb[512 * rr] = acc[rr];
    }
    __syncthreads();

    // ---- S3 + look-ahead.  Sub-block (s, jb), jb > kb, of the array is E's row block s for s <= kb (upper
    // part) and M's for s >= jb (lower part).  Wave 0 updates the NEXT diagonal sub-block first and
    // factorises it (S1 of step kb+1: a 16-step dependent chain on one wave) while waves 1-3 share the
    // other sub-blocks of this step, so the serial sweep hides behind the MFMA work.
    if (wave == 0) {
      if (kb < 7) {
        s3(kb, kb + 1, kb + 1, 0, 0, false);
        s1(kb + 1);
      }
    } else {
      int cnt = 0, ps = -1, pj = -1;
      for (int jb = kb + 1; jb < 8; ++jb)
        for (int sb = 0; sb < 8; ++sb) {
          if (!(sb <= kb || sb >= jb) || (sb == kb + 1 && jb == kb + 1)) continue;
          if (cnt++ % 3 != wave - 1) continue;
          if (ps < 0) { ps = sb; pj = jb; continue; }
          s3(kb, ps, pj, sb, jb, true);
          ps = -1;
        }
      if (ps >= 0) s3(kb, ps, pj, ps, pj, false);
    }
    __syncthreads();
  }
  bad = flag[0];

  // ---- log det partial and status
  double lsum = 0.0;
  if (!bad && tid < kTile) lsum = log(dvec[tid]);
  for (int off = 32; off > 0; off >>= 1) lsum += __shfl_down(lsum, off, 64);
  if (lane == 0) red[wave] = lsum;
  __syncthreads();
  if (tid == 0) {
    g.logdet_part[(size_t)b * g.nt + g.j] = bad ? kNaN : (red[0] + red[1] + red[2] + red[3]);
    if (bad && g.status[b] == 0) g.status[b] = g.j * kTile + bad;
  }

  // ---- L back to the matrix (lower sub-blocks; zeros above the diagonal inside the diagonal ones)
  {
    const int r = tid & 127, ch = tid >> 7;
#pragma unroll 4
    for (int c2 = 0; c2 < kTile; c2 += 2) {
      const int c = c2 + ch;
      if ((r >> 4) >= (c >> 4)) C[r + (size_t)c * ld] = bad ? kNaN : Ms[idx(r, c)];
    }
  }
  // ---- W = L^-1 = E': W[c + t*128] = E[t][c].  E is read along its rows (conflict-free) and transposed
  // on the way out by an MFMA against the identity, so that the stores run along c.
  double* W = g.invd + (size_t)b * g.invd_stride + (size_t)g.j * kTile * kTile;
  for (int u = wave; u < 64; u += 4) {
    const int tb = u >> 3, cb = u & 7;
    d4 acc = d4{0.0, 0.0, 0.0, 0.0};
    if (cb >= tb && !bad) {
      const double* ep = cb == tb ? ed_frag + tb * 256 : blk(tb, cb);   // E[t = l15][c = 4 kk + l4]
      const int estep = cb == tb ? 64 : 512;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const double idf = (4 * kk + l4) == l15 ? 1.0 : 0.0;            // I[k = l4 (+ 4 kk)][j = l15]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ep[estep * kk], idf, acc, 0, 0, 0);   // D[i = t][j = c]
      }
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
      W[(16 * cb + l15) + (size_t)(16 * tb + l4 + 4 * rr) * kTile] = bad ? kNaN : acc[rr];
  }
}

