// Batched blocked Cholesky + solves for n > 128 (BASELINE config 4: n = 4096).
//
// LEFT-looking, 128-wide block columns, all matrices of a chunk in one launch per phase.  For block column j:
//   update : T_ij = A_ij - sum_{k<j} L_ik L_jk'      all tiles i >= j   (f64 MFMA GEMM)
//            + for i = j, in the same workgroup: the right-hand-side rows, then L_jj = chol(T_jj), W_j = L_jj^-1
//   trsm   : L_ij = T_ij W_j'                        tiles i > j        (f64 MFMA GEMM)
// The right-hand sides [y 1] are rows npad, npad+1 of an (npad+128) x npad array per matrix: the update treats
// them with the diagonal tile (diag_rhs_tile), trsm as a thin tile row, so when the sweep ends they hold
// Z' = [y 1]' L^-T, i.e. the forward substitution L z = b is done -- no separate solve pass over the 4n^2 B
// factor.  A last tiny kernel turns Z and the pivots into what the reference's dmnorm / beta.MLE return
// (HX:458-460, HX:570).  Left-looking because every tile is then WRITTEN once (the right-looking form rewrites
// the whole trailing matrix per block column).  The reads stay: tile (i, j) streams its own row panel
// L_i,0..j-1 from HBM and shares the column panel L_j,0..j-1 with the other tiles of its matrix through that
// XCD's L2, so the algorithmic traffic is sum_j (nt - j) (j 128 KiB + 256 KiB) = 0.85 GB per n = 4096 matrix,
// 1.75 GB per launch of 64 matrices averaged over the sweep; measured 1.93 GB (PMC, profiles/r02n/pmc_traffic64.json).
// At 128-wide tiles that is 16 flop per HBM byte against a machine balance of ~10: MFMA-bound, with HBM at a
// third of its peak.
//
// Two generations of k-loop live here.  Round 3 (tile_accumulate_il for whole update tiles, diag_rhs_accumulate<W> for
// the diagonal workgroup): ds_read_b128 fragments on row-interleaved sub-tiles, buffer_load ... lds stage requests,
// one request per MFMA -- 75 TFLOP/s for the bare loop against 67 - 71 (tests/hip/update_loop_probe.hip).  Round 2
// (gemm_accumulate, strip_accumulate_ring): the panel solve, the strips, R^-1 tiles, and the fallback of the new
// loops when a panel spans 4 GiB or more.  Every loop sums a tile element's k four at a time in ascending order,
// so they are interchangeable bit for bit.
//
// Schedule of an update launch (profiles/r02_update_schedule.md): a launch of W workgroups takes ceil(W / 256)
// steps -- one workgroup per CU (one wave per SIMD) already keeps the four MFMA pipes busy (84 % in round 2, 97 % of
// the bare-MFMA rate now), the second resident workgroup only fills bubbles.  The diagonal workgroups (one per matrix: lower triangle + right-hand sides, then the
// block factorisation) are dispatched first; whole tiles follow; the <= 128 tiles of a partial last step run as two
// ring-pipelined half-width strips each.
//
// MFMA: v_mfma_f64_16x16x4_f64.  Operand lane map (one f64 per lane):
//   A[i = lane&15][k = lane>>4],  B[k = lane>>4][j = lane&15],
//   D[i = (lane>>4) + 4*r][j = lane&15] for accumulator register r = 0..3.
// We feed A := Q (the block-column operand) and B := P (the block-row operand), so D's
// lane index runs along matrix ROWS: a wave's store is 4 columns x 128 contiguous bytes.
#include <cstdlib>
#include <vector>

#include "ccgp_internal.h"
#include "sched_logic.h"

namespace ccgp {

typedef double d4 __attribute__((ext_vector_type(4)));

namespace {

// threadIdx.x as the tile code sees it: behind an asm the compiler cannot look through, so that nothing derived from it is
// hoisted out of the persistent loop of chol_sched_kernel and kept live across every tile variant (which cost 256 VGPRs and
// 416 B of scratch per lane); each variant recomputes its few lane offsets per tile instead.
__device__ __forceinline__ int tid_now() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}


struct GemmArgs {
  double* A;
  size_t a_stride;
  int npad;
  double* invd;
  size_t invd_stride;
  int j, nt, nb;
  int ld;    // leading dimension = npad + 128 (right-hand-side tile row) + 128 * ne
  int ne;    // extra full tile rows below the right-hand-side row (cross-correlation rows)
  int extra_lower;  // 1: the extra rows are the identity (block row te is zero left of block column te)
  int mode;  // 0: update, 1: trsm
  // prediction from a KEPT factor (ccgp_predict_from_factorset): the extra tile rows live in their own
  // buffer E (ne x 128 rows, leading dimension lde, e_stride elements per matrix) instead of below the
  // matrix, and a launch covers ONLY those rows (rows_only): the factor itself is read-only
  double* E;
  size_t e_stride;
  int lde;
  int rows_only;
  // update only: the workgroup that has just finished the diagonal tile goes on to factorise and invert it
  // (diag_factor) while the other tiles of the launch are still being updated
  int fuse_diag;
  int tail_strips;   // host switch (CCGP_OPT_TAIL_STRIPS)
  int wide;          // host switch (CCGP_OPT_WIDE_OFFSETS): 64-bit-pointer loops everywhere
  int n_s1;   // update, S = 1 kernel: tiles [0, n_s1) (dispatch order) run whole, the rest as tail_s ring-pipelined strips
  int tail_s; // 2 (half-width) or 4 (quarter-width) strips per tail tile
  double* logdet_part;
  int* status;
  int n;
  double ptol;   // pivot_tolerance(mean_mode, n) for the fused diagonal factorisation
};

// One output tile strip, C = C - P Q' (MODE 0, update) or C = P Q' (MODE 1, trsm): 128 rows x
// (128 / S) columns per workgroup, K-loop over 16-deep double-buffered LDS stages.
//   S = 1: 2 x 2 waves of 64 x 64      S = 2: 2 x 2 waves of 64 x 32
// S = 2 re-reads the P panel twice through L2; it serves the rows-only sweeps of a kept factor (few, long rows)
// and, in its ring-pipelined form (strip_accumulate_ring), the partial last step of an update launch.
// THIN = trsm's right-hand-side tile row: only its first 16 rows carry data, so the waves split
// the strip's columns between them (16 rows x 32 columns per wave at S = 1).
//
// Staging is LDS-DMA: the next stage is filled by global_load_lds (16 B per lane, one
// wave-instruction = 1 KiB landing linearly in LDS) while the MFMAs of the current stage run,
// so there is no VGPR staging and no ds_write pass (measured 3 % faster than register staging,
// profiles/r01c_strip_selection.md).  Because the DMA destination is lane-linear, the LDS image
// is UNPADDED ([k][128] / [k][CW] doubles) and bank conflicts are removed by an XOR swizzle
// instead: the 16-row block index of column k is XORed with (k & 1), applied on the per-lane
// SOURCE address and again on the fragment read (both sides or neither).
template <int S, bool THIN>
struct TileGeom {
  static constexpr int CW = kTile / S;                                       // columns per workgroup
  static constexpr int NX = THIN ? (S == 1 ? 2 : 1) : (S == 1 ? 4 : 2);      // 16-wide column sub-tiles per wave
  static constexpr int NY = THIN ? 1 : (S == 4 ? 2 : 4);                     // 16-high row sub-tiles per wave
  static constexpr int BK = 16;   // k-depth of one LDS stage (32 was measured 3 % slower in round 1)
  __device__ static int row0(int wave) { return THIN ? 0 : (S == 4 ? wave * 32 : (wave >> 1) * 64); }
  __device__ static int col0(int wave) {
    return THIN ? wave * NX * 16 : (S == 4 ? 0 : (wave & 1) * (S == 1 ? 64 : 32));
  }
};

// acc[x][y] += (Q strip)(P tile)' over Kdim: accumulator register r of sub-tile (x, y) is element
// (row0 + 16y + lane&15, col0 + 16x + (lane>>4) + 4r) of the 128 x CW output.
// TRI (trsm only, S = 1): Q = W_j is LOWER triangular (W[c][k] = 0 for k > c), so the 16-column
// sub-tile cb contributes nothing once the stage index kt exceeds cb; those MFMAs are skipped,
// and the sub-tiles are dealt to the two wave columns interleaved (cb = wn + 2x) so that both
// keep a similar share of the surviving work (20 and 16 of 32 units instead of 26 and 10).
template <int S, bool THIN, bool TRI>
__device__ __forceinline__ int col_block(int wave, int x) {
  if constexpr (TRI && !THIN) return (wave & 1) + 2 * x;
  else return (TileGeom<S, THIN>::col0(wave) >> 4) + x;
}

template <int S, bool THIN, bool TRI = false>
__device__ __forceinline__ void gemm_accumulate(double* smem, const double* P, int ldP, const double* Q,
                                                int ldQ, int Kdim,
                                                d4 (&acc)[TileGeom<S, THIN>::NX][TileGeom<S, THIN>::NY]) {
  static_assert(!TRI || S == 1, "the triangular skip is only wired for full-width tiles");
  constexpr int CW = TileGeom<S, THIN>::CW;
  constexpr int NX = TileGeom<S, THIN>::NX;
  constexpr int NY = TileGeom<S, THIN>::NY;
  constexpr int BKs = TileGeom<S, THIN>::BK;
  constexpr int STAGE = BKs * kTile + BKs * CW;        // doubles per stage, unpadded
  constexpr int CPI = kTile / CW;                      // Q columns covered by one wave-instruction (1, 2, 4)
  constexpr int QI = BKs / CPI / 4;                    // Q wave-instructions per wave per stage (4, 2, 1)
  const int tid = tid_now(), lane = tid & 63, wave = tid >> 6;
  const int row0 = TileGeom<S, THIN>::row0(wave);
  const int col0 = TileGeom<S, THIN>::col0(wave);
  const bool active = !THIN || col0 < CW;
  const int l15 = lane & 15, l4 = lane >> 4;

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
    double* Qs_ = smem + stage * STAGE + BKs * kTile + wave * kTile;
#pragma unroll
    for (int q = 0; q < BKs / 4; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pP + q * stepP),
                                       (__attribute__((address_space(3))) void*)(Ps_ + 4 * q * kTile), 16, 0, 0);
#pragma unroll
    for (int q = 0; q < QI; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pQ + q * stepQ),
                                       (__attribute__((address_space(3))) void*)(Qs_ + 4 * q * kTile), 16, 0, 0);
    pP += (BKs / 4) * stepP;
    pQ += QI * stepQ;
  };

  // Software pipeline.  Left to itself hipcc issues each k-step's fragment ds_reads and waits
  // lgkmcnt(0) right before that step's MFMAs, exposing the LDS latency once per k-step.  Here the
  // fragments of step kk+1 are loaded into a second register set BEFORE the MFMAs of step kk, and
  // the stage barrier sits in front of the LAST k-step: the next stage's DMA and its first
  // fragments are issued right after the barrier and land underneath that last k-step's MFMAs.
  // (sched_barrier pins the order; the waits are the compiler's.)
  static_assert(BKs == 16, "the pipeline below is written for four k-steps per stage");
  double pfA[NY], qfA[NX], pfB[NY], qfB[NX];
  const int sw = l4 & 1;   // k parity of this lane's fragment element (k = 4 kk + l4)
#define CCGP_LOADF(PF, QF, STG, KK)                                                              \
  do {                                                                                          \
    const double* Ps_ = smem + (STG) * STAGE;                                                   \
    const double* Qs_ = Ps_ + BKs * kTile;                                                      \
    _Pragma("unroll") for (int y = 0; y < NY; ++y)                                              \
        PF[y] = Ps_[((KK) * 4 + l4) * kTile + ((((row0 >> 4) + y) ^ sw) << 4) + l15];           \
    _Pragma("unroll") for (int x = 0; x < NX; ++x)                                              \
        QF[x] = Qs_[((KK) * 4 + l4) * CW + ((col_block<S, THIN, TRI>(wave, x) ^ sw) << 4) + l15]; \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  } while (0)
// MFMAs of one k-step in two parts, so that the NEXT step's fragment loads can be issued after
// the first part: hipcc only emits lgkmcnt(0) waits (it does not count across the loop), and a
// load issued right before a wait would expose its full latency; issued here it has the rest of
// this step's MFMAs (12 of 16 at S = 1) to land before the next wait.
#define CCGP_MFMAS(PF, QF, KT, X0, X1)                                                           \
  do {                                                                                          \
    if (active) {                                                                               \
      _Pragma("unroll") for (int x = (X0); x < (X1); ++x) {                                     \
        if (TRI && col_block<S, THIN, TRI>(wave, x) * 16 + 15 < (KT) * BKs) continue;           \
        _Pragma("unroll") for (int y = 0; y < NY; ++y)                                          \
            acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(QF[x], PF[y], acc[x][y], 0, 0, 0); \
      }                                                                                         \
    }                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  } while (0)
  constexpr int XH = NX > 1 ? 1 : NX;   // MFMAs x < XH go before the prefetch, the rest after

  const int nk = Kdim / BKs;
  issue(0);
  __syncthreads();   // drains the DMA (vmcnt(0)) and publishes it
  if (nk > 1) issue(1);
  CCGP_LOADF(pfA, qfA, 0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int stg = kt & 1;
    CCGP_MFMAS(pfA, qfA, kt, 0, XH);
    CCGP_LOADF(pfB, qfB, stg, 1);
    CCGP_MFMAS(pfA, qfA, kt, XH, NX);
    CCGP_MFMAS(pfB, qfB, kt, 0, XH);
    CCGP_LOADF(pfA, qfA, stg, 2);
    CCGP_MFMAS(pfB, qfB, kt, XH, NX);
    CCGP_MFMAS(pfA, qfA, kt, 0, XH);
    CCGP_LOADF(pfB, qfB, stg, 3);
    CCGP_MFMAS(pfA, qfA, kt, XH, NX);
    if (kt + 1 < nk) {
      __syncthreads();                       // stage kt fully read by everyone, stage kt+1 landed
      if (kt + 2 < nk) issue(stg);           // refill the buffer that was just released
    }
    CCGP_MFMAS(pfB, qfB, kt, 0, XH);         // last k-step of stage kt, from registers
    if (kt + 1 < nk) CCGP_LOADF(pfA, qfA, stg ^ 1, 0);   // first fragments of the next stage
    CCGP_MFMAS(pfB, qfB, kt, XH, NX);
  }
#undef CCGP_LOADF
#undef CCGP_MFMAS
}



// buffer_load ... lds addresses a panel as base + 32-bit offset: rows k < Kdim (+ the 16 of a stage in flight),
// leading dimension ld, 8-byte elements, plus a lane's 1 KiB
__device__ __forceinline__ bool fits_buffer_offsets(int Kdim, int ld) {
  return ((size_t)Kdim + 32) * (size_t)ld * 8 < 0xFFFF0000ull;
}

// ---- whole update tile, round-3 loop ("interleaved") ----------------------------------------------------
// tests/hip/update_loop_probe.hip rebuilt the loop above with its ingredients switchable and found where the 15 % of
// non-MFMA cycles go (MI355X, steady state, no epilogue): bare MFMAs 77.5 TFLOP/s; + the 32 ds_read_b64 fragment
// loads of a stage 71 - 74; + the eight global_load_lds with their per-lane 64-bit address arithmetic 67 - 71.  Three
// changes bring the full loop back to 75.5 - 76.5:
//   * the four 16 x 16 sub-tiles of a wave interleave their rows (sub-tile y holds rows row0 + 4 l15 + y, sub-tile x
//     columns col0 + 4 i + x), so a lane's four P (or Q) fragments of a k-step are 32 CONTIGUOUS bytes of the
//     linear LDS image: two ds_read_b128 instead of four ds_read_b64, conflict-free without the XOR swizzle;
//   * the stage requests are buffer_load ... lds: the lane offset is ONE constant 32-bit VGPR, rows and stages
//     advance in SGPRs (soffset) -- no vector instruction between the MFMAs;
//   * fragment reads and stage requests are dealt out one per MFMA (an MFMA holds its pipe for 64 cycles during
//     which the wave issues the next request for free) instead of in bursts.
// Every output element still accumulates its k-sum four at a time in ascending order: the bits are those of
// gemm_accumulate, whatever the sub-tile a row or column lands in (ring strips, trsm and the diagonal workgroup
// keep the older loop).
//   acc[x][y] register r  =  element (row0 + 4 l15 + y,  col0 + 4 (l4 + 4 r) + x)  of the 128 x 128 tile.
__device__ __forceinline__ void tile_accumulate_il(double* smem, const double* P, int ldP, const double* Q,
                                                   int ldQ, int Kdim, d4 (&acc)[4][4]) {
  constexpr int BKs = 16, STAGE = 2 * BKs * kTile;     // doubles per stage: P image [16][128], then Q image [16][128]
  typedef double d2 __attribute__((ext_vector_type(2)));
  const int tid = tid_now(), lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = (wave >> 1) * 64, col0 = (wave & 1) * 64;
  const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = d4{0.0, 0.0, 0.0, 0.0};

  // wave w requests rows k = w + 4 q of either image: one wave-instruction = one 1 KiB row, lane-linear
  const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)P, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc((void*)Q, 0, -1, 0x00020000);
  const unsigned lane_b = (unsigned)lane * 16;
  const unsigned stepP = (unsigned)ldP * 32, stepQ = (unsigned)ldQ * 32;   // four rows, bytes
  unsigned oP = (unsigned)wave * (unsigned)ldP * 8, oQ = (unsigned)wave * (unsigned)ldQ * 8;
  auto request = [&](int stage, int r) {   // r = 0..3: P rows, 4..7: Q rows
    double* dst = smem + stage * STAGE + (r >> 2) * (BKs * kTile) + (wave + 4 * (r & 3)) * kTile;
    if (r < 4)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rP, (__attribute__((address_space(3))) void*)dst, 16, lane_b,
                                               oP + (unsigned)(r & 3) * stepP, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rQ, (__attribute__((address_space(3))) void*)dst, 16, lane_b,
                                               oQ + (unsigned)(r & 3) * stepQ, 0, 0);
  };
  auto advance = [&]() { oP += 4 * stepP; oQ += 4 * stepQ; };

  const int pofs = l4 * kTile + row0 + 4 * l15, qofs = BKs * kTile + l4 * kTile + col0 + 4 * l15;
  double pfA[4], qfA[4], pfB[4], qfB[4];
#define CCGP_SB __builtin_amdgcn_sched_barrier(0)
#define CCGP_ILOAD(PF, QF, STG, KK, I)                                                            \
  do {                                                                                           \
    const double* St_ = smem + (STG) * STAGE + (KK) * 4 * kTile;                                 \
    if ((I) == 0) { const d2 v_ = *(const d2*)(St_ + pofs); PF[0] = v_[0]; PF[1] = v_[1]; }      \
    if ((I) == 1) { const d2 v_ = *(const d2*)(St_ + pofs + 2); PF[2] = v_[0]; PF[3] = v_[1]; }  \
    if ((I) == 2) { const d2 v_ = *(const d2*)(St_ + qofs); QF[0] = v_[0]; QF[1] = v_[1]; }      \
    if ((I) == 3) { const d2 v_ = *(const d2*)(St_ + qofs + 2); QF[2] = v_[0]; QF[3] = v_[1]; }  \
  } while (0)
// one k-step: 16 MFMAs; behind MFMA i < 4 the i-th fragment read of the NEXT k-step (LD), behind MFMA i < 8 of a
// stage's last k-step the i-th request of the stage after next (RQ)
#define CCGP_ISTEP(PFc, QFc, PFn, QFn, STGn, KKn, LD, RQ, STGr)                                   \
  do {                                                                                           \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                             \
      acc[i >> 2][i & 3] =                                                                       \
          __builtin_amdgcn_mfma_f64_16x16x4f64(QFc[i >> 2], PFc[i & 3], acc[i >> 2][i & 3], 0, 0, 0); \
      if ((LD) && i < 4) CCGP_ILOAD(PFn, QFn, STGn, KKn, i);                                     \
      if ((RQ) && i < 8) request(STGr, i);                                                       \
      CCGP_SB;                                                                                   \
    }                                                                                            \
  } while (0)
#define CCGP_ISTAGE(MORE, REFILL)                                                                 \
  do {                                                                                           \
    const int stg = kt & 1;                                                                      \
    CCGP_ISTEP(pfA, qfA, pfB, qfB, stg, 1, true, false, 0);                                      \
    CCGP_ISTEP(pfB, qfB, pfA, qfA, stg, 2, true, false, 0);                                      \
    CCGP_ISTEP(pfA, qfA, pfB, qfB, stg, 3, true, false, 0);                                      \
    if (MORE) {                                                                                  \
      __syncthreads();          /* stage kt read by everyone (its last fragments are in registers), kt+1 landed */ \
      CCGP_SB;                                                                                   \
    }                                                                                            \
    CCGP_ISTEP(pfB, qfB, pfA, qfA, stg ^ 1, 0, MORE, REFILL, stg);                               \
    if (REFILL) advance();                                                                       \
  } while (0)

  const int nk = Kdim / BKs;
#pragma unroll
  for (int r = 0; r < 8; ++r) request(0, r);
  advance();
  __syncthreads();   // drains the requests (vmcnt(0)) and publishes the stage
  if (nk > 1) {
#pragma unroll
    for (int r = 0; r < 8; ++r) request(1, r);
    advance();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) CCGP_ILOAD(pfA, qfA, 0, 0, i);
  CCGP_SB;
  int kt = 0;
  for (; kt < nk - 2; ++kt) CCGP_ISTAGE(true, true);
  if (nk >= 2) { CCGP_ISTAGE(true, false); ++kt; }
  CCGP_ISTAGE(false, false);
#undef CCGP_ISTAGE
#undef CCGP_ISTEP
#undef CCGP_ILOAD
#undef CCGP_SB
}

// C -= acc for a whole update tile in the interleaved sub-tile map: a lane holds four CONTIGUOUS rows of each of
// its 16 columns, so the read-modify-write is 16 + 16 32-byte accesses per lane (64 + 64 of 8 bytes before)
__device__ __forceinline__ void update_tile_il(double* smem, const double* P, int ldP, const double* Q, int ldQ,
                                               int Kdim, double* C, int ld) {
  d4 acc[4][4];
  tile_accumulate_il(smem, P, ldP, Q, ldQ, Kdim, acc);
  const int lane = tid_now() & 63, wave = tid_now() >> 6;
  const int row0 = (wave >> 1) * 64, col0 = (wave & 1) * 64;
  const int l15 = lane & 15, l4 = lane >> 4;
  double* Cl = C + row0 + 4 * l15 + (size_t)(col0 + 4 * l4) * ld;
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    d4 cv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) cv[r] = *(const d4*)(Cl + (size_t)(16 * r + x) * ld);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      d4 o;
#pragma unroll
      for (int y = 0; y < 4; ++y) o[y] = cv[r][y] - acc[x][y][r];
      *(d4*)(Cl + (size_t)(16 * r + x) * ld) = o;
    }
  }
}

// ---- half-width strip with a four-stage ring (tail of an update launch) ---------------------------
// Same strip geometry, fragments and per-accumulator MFMA order as gemm_accumulate<2, false> (so the same
// bits), but the LDS holds FOUR stages of 8 k-columns and a stage is requested THREE stages ahead, with counted
// waits (s_waitcnt vmcnt(6 | 3 | 0): three DMA instructions per wave and stage) instead of a drain.  A strip does
// half the MFMAs of a full tile per stage, so ALONE on a CU -- which is where the tail strips run -- the
// one-stage-ahead loop is bound by the latency of its own requests (13.7 us per 128-deep block against 16.3 for
// a whole tile, profiles/r02_update_schedule.md section 3); for full tiles the same ring was measured and
// rejected (section 7: they are MFMA-bound even alone).
__device__ __forceinline__ void strip_accumulate_ring(double* smem, const double* P, int ldP, const double* Q,
                                                      int ldQ, int Kdim, d4 (&acc)[2][4]) {
  constexpr int BKd = 8, NST = 4, CW = kTile / 2;
  constexpr int STAGE = BKd * kTile + BKd * CW;   // doubles: P image [8][128], then Q image [8][64]
  const int tid = tid_now(), lane = tid & 63, wave = tid >> 6;
  const int row0 = (wave >> 1) * 64, col0 = (wave & 1) * 32;
  const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = d4{0.0, 0.0, 0.0, 0.0};

  // P: wave w stages columns k = w and w + 4 (parity w & 1).  Q: one instruction per wave covers columns
  // 2w (lanes 0-31) and 2w + 1 (lanes 32-63): parity = lane >> 5
  const int psrc = ((((lane >> 3) ^ (wave & 1)) << 4) + ((lane & 7) << 1));
  const double* pP = P + psrc + (size_t)wave * ldP;
  const int ql = lane & 31, qc = lane >> 5;
  const int qsrc = ((((ql >> 3) ^ qc) << 4) + ((ql & 7) << 1));
  const double* pQ = Q + qsrc + (size_t)(2 * wave + qc) * ldQ;
  auto issue = [&](int slot) {
    double* Ps_ = smem + slot * STAGE + wave * kTile;
    double* Qs_ = smem + slot * STAGE + BKd * kTile + wave * 2 * CW;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pP,
                                     (__attribute__((address_space(3))) void*)Ps_, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pP + (size_t)4 * ldP),
                                     (__attribute__((address_space(3))) void*)(Ps_ + 4 * kTile), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pQ,
                                     (__attribute__((address_space(3))) void*)Qs_, 16, 0, 0);
    pP += (size_t)BKd * ldP;
    pQ += (size_t)BKd * ldQ;
  };
  // s_waitcnt simm16 on gfx9: vmcnt[3:0] | expcnt << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14; expcnt 7 = no wait
  auto wait_and_meet = [&](int later_stages) {   // wave-uniform: stages requested after the one that must have landed
    if (later_stages >= 3) __builtin_amdgcn_s_waitcnt(0x0079);        // vmcnt(9) lgkmcnt(0)
    else if (later_stages == 2) __builtin_amdgcn_s_waitcnt(0x0076);   // vmcnt(6) lgkmcnt(0)
    else if (later_stages == 1) __builtin_amdgcn_s_waitcnt(0x0073);   // vmcnt(3) lgkmcnt(0)
    else __builtin_amdgcn_s_waitcnt(0x0070);                          // vmcnt(0) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
  };

  const int sw = l4 & 1;
  const int offP = l4 * kTile + l15, offQ = BKd * kTile + l4 * CW + l15;
  int rbo[4], cbo[2];
#pragma unroll
  for (int t = 0; t < 4; ++t) rbo[t] = (((row0 >> 4) + t) ^ sw) << 4;
#pragma unroll
  for (int t = 0; t < 2; ++t) cbo[t] = (((col0 >> 4) + t) ^ sw) << 4;
  double pfA[4], qfA[2], pfB[4], qfB[2];
#define CCGP_RLOADF(PF, QF, SLOT, KK)                                                            \
  do {                                                                                          \
    const double* St_ = smem + (SLOT) * STAGE;                                                  \
    _Pragma("unroll") for (int y = 0; y < 4; ++y) PF[y] = St_[(KK) * 4 * kTile + offP + rbo[y]]; \
    _Pragma("unroll") for (int x = 0; x < 2; ++x) QF[x] = St_[(KK) * 4 * CW + offQ + cbo[x]];   \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  } while (0)
#define CCGP_RMFMAS(PF, QF, X0, X1)                                                              \
  do {                                                                                          \
    _Pragma("unroll") for (int x = (X0); x < (X1); ++x)                                         \
      _Pragma("unroll") for (int y = 0; y < 4; ++y)                                             \
          acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(QF[x], PF[y], acc[x][y], 0, 0, 0);   \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  } while (0)

  const int nk = Kdim / BKd;
  const int pre = nk < NST ? nk : NST;
  for (int t = 0; t < pre; ++t) issue(t);
  __builtin_amdgcn_sched_barrier(0);
  wait_and_meet(pre - 1);                               // stage 0 landed; the later ones stay in flight
  CCGP_RLOADF(pfA, qfA, 0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int slot = kt & (NST - 1);
    CCGP_RLOADF(pfB, qfB, slot, 1);
    CCGP_RMFMAS(pfA, qfA, 0, 2);                        // k-step 0 of stage kt
    if (kt + 1 < nk) {
      // stage kt+1 landed for every wave, and stage kt has been read by every wave (its last fragments are in
      // registers: lgkmcnt(0)), so its slot takes stage kt+4
      const int last_issued = kt + NST - 1 < nk - 1 ? kt + NST - 1 : nk - 1;
      wait_and_meet(last_issued - (kt + 1));
      if (kt + NST < nk) issue(slot);
      __builtin_amdgcn_sched_barrier(0);
    }
    CCGP_RMFMAS(pfB, qfB, 0, 1);                        // k-step 1 of stage kt, from registers
    if (kt + 1 < nk) CCGP_RLOADF(pfA, qfA, (kt + 1) & (NST - 1), 0);
    CCGP_RMFMAS(pfB, qfB, 1, 2);
  }
#undef CCGP_RLOADF
#undef CCGP_RMFMAS
}

// ---- quarter-width strip on the same ring (round 4) ------------------------------------------------------
// 128 rows x 32 columns per workgroup in TileGeom<4>'s layout (wave w: rows 32 w .. 32 w + 31, all 32 columns: 2 x 2 sub-tiles,
// four MFMAs per k-step), for the partial last step of an update launch whose tail is a QUARTER of a step (64 tiles on 256
// CUs: 256 quarter strips fill the chip where 128 half strips filled half of it) or half a step (512 quarter strips, two per
// CU).  Fragments, swizzle and per-accumulator MFMA order are gemm_accumulate<4, false>'s: same bits.  The Q image of a stage
// is 8 k-rows of 256 bytes: two wave-instructions; waves 2 and 3 request the same rows again (same bytes to the same place)
// so that every wave counts three requests per stage and the counted waits stay wave-uniform.
__device__ __forceinline__ void strip_accumulate_ring4(double* smem, const double* P, int ldP, const double* Q,
                                                       int ldQ, int Kdim, d4 (&acc)[2][2]) {
  constexpr int BKd = 8, NST = 4, CW = kTile / 4;
  constexpr int STAGE = BKd * kTile + BKd * CW;   // doubles: P image [8][128], then Q image [8][32]
  const int tid = tid_now(), lane = tid & 63, wave = tid >> 6;
  const int row0 = wave * 32;
  const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) acc[x][y] = d4{0.0, 0.0, 0.0, 0.0};
  const int psrc = ((((lane >> 3) ^ (wave & 1)) << 4) + ((lane & 7) << 1));
  const double* pP = P + psrc + (size_t)wave * ldP;
  // Q: one wave-instruction = k-rows 4 (w & 1) + (lane >> 4), 16 lanes x 16 bytes each; parity of the row = (lane >> 4) & 1
  const int qrk = lane >> 4, ql = lane & 15;
  const int qsrc = ((((ql >> 3) ^ (qrk & 1)) << 4) + ((ql & 7) << 1));
  const double* pQ = Q + qsrc + (size_t)(4 * (wave & 1) + qrk) * ldQ;
  auto issue = [&](int slot) {
    double* Ps_ = smem + slot * STAGE + wave * kTile;
    double* Qs_ = smem + slot * STAGE + BKd * kTile + (wave & 1) * 4 * CW;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pP,
                                     (__attribute__((address_space(3))) void*)Ps_, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pP + (size_t)4 * ldP),
                                     (__attribute__((address_space(3))) void*)(Ps_ + 4 * kTile), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pQ,
                                     (__attribute__((address_space(3))) void*)Qs_, 16, 0, 0);
    pP += (size_t)BKd * ldP;
    pQ += (size_t)BKd * ldQ;
  };
  auto wait_and_meet = [&](int later_stages) {   // as in strip_accumulate_ring: three requests per wave and stage
    if (later_stages >= 3) __builtin_amdgcn_s_waitcnt(0x0079);
    else if (later_stages == 2) __builtin_amdgcn_s_waitcnt(0x0076);
    else if (later_stages == 1) __builtin_amdgcn_s_waitcnt(0x0073);
    else __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
  };
  const int sw = l4 & 1;
  const int offP = l4 * kTile + l15, offQ = BKd * kTile + l4 * CW + l15;
  int rbo[2], cbo[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    rbo[t] = (((row0 >> 4) + t) ^ sw) << 4;
    cbo[t] = (t ^ sw) << 4;
  }
  double pfA[2], qfA[2], pfB[2], qfB[2];
#define CCGP_QLOADF(PF, QF, SLOT, KK)                                                            \
  do {                                                                                          \
    const double* St_ = smem + (SLOT) * STAGE;                                                  \
    _Pragma("unroll") for (int y = 0; y < 2; ++y) PF[y] = St_[(KK) * 4 * kTile + offP + rbo[y]]; \
    _Pragma("unroll") for (int x = 0; x < 2; ++x) QF[x] = St_[(KK) * 4 * CW + offQ + cbo[x]];   \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  } while (0)
#define CCGP_QMFMAS(PF, QF, X0, X1)                                                              \
  do {                                                                                          \
    _Pragma("unroll") for (int x = (X0); x < (X1); ++x)                                         \
      _Pragma("unroll") for (int y = 0; y < 2; ++y)                                             \
          acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(QF[x], PF[y], acc[x][y], 0, 0, 0);   \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  } while (0)
  const int nk = Kdim / BKd;
  const int pre = nk < NST ? nk : NST;
  for (int t = 0; t < pre; ++t) issue(t);
  __builtin_amdgcn_sched_barrier(0);
  wait_and_meet(pre - 1);
  CCGP_QLOADF(pfA, qfA, 0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int slot = kt & (NST - 1);
    CCGP_QLOADF(pfB, qfB, slot, 1);
    CCGP_QMFMAS(pfA, qfA, 0, 2);
    if (kt + 1 < nk) {
      const int last_issued = kt + NST - 1 < nk - 1 ? kt + NST - 1 : nk - 1;
      wait_and_meet(last_issued - (kt + 1));
      if (kt + NST < nk) issue(slot);
      __builtin_amdgcn_sched_barrier(0);
    }
    CCGP_QMFMAS(pfB, qfB, 0, 1);
    if (kt + 1 < nk) CCGP_QLOADF(pfA, qfA, (kt + 1) & (NST - 1), 0);
    CCGP_QMFMAS(pfB, qfB, 1, 2);
  }
#undef CCGP_QLOADF
#undef CCGP_QMFMAS
}

// C = C - acc (mode 0) or C = acc (mode 1) for one strip.
template <int S, bool THIN, bool TRI = false, bool RING = false>
__device__ __forceinline__ void gemm_tile(double* smem, const double* P, int ldP, const double* Q,
                                          int ldQ, int Kdim, double* C, int ld, int mode) {
  constexpr int NX = TileGeom<S, THIN>::NX;
  constexpr int NY = TileGeom<S, THIN>::NY;
  d4 acc[NX][NY];
  if constexpr (RING) {
    static_assert((S == 2 || S == 4) && !THIN && !TRI, "the four-stage ring exists for half- and quarter-width update strips");
    if constexpr (S == 2) strip_accumulate_ring(smem, P, ldP, Q, ldQ, Kdim, acc);
    else strip_accumulate_ring4(smem, P, ldP, Q, ldQ, Kdim, acc);
  } else {
    gemm_accumulate<S, THIN, TRI>(smem, P, ldP, Q, ldQ, Kdim, acc);
  }
  const int lane = tid_now() & 63, wave = tid_now() >> 6;
  const int row0 = TileGeom<S, THIN>::row0(wave), col0 = TileGeom<S, THIN>::col0(wave);
  const int l15 = lane & 15, l4 = lane >> 4;
  if (THIN && col0 >= TileGeom<S, THIN>::CW) return;
  // all loads of one 16-column group are issued before the first store: written as
  // `*p = *p - acc` the compiler orders every load behind the previous store and the epilogue
  // becomes 16-64 serial memory round trips per thread
#pragma unroll
  for (int x = 0; x < NX; ++x) {
    double cv[NY][4];
    if (mode == 0) {
#pragma unroll
      for (int y = 0; y < NY; ++y)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          cv[y][r] = C[row0 + y * 16 + l15 + (size_t)(col_block<S, THIN, TRI>(wave, x) * 16 + l4 + 4 * r) * ld];
    }
#pragma unroll
    for (int y = 0; y < NY; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double* p = C + row0 + y * 16 + l15 + (size_t)(col_block<S, THIN, TRI>(wave, x) * 16 + l4 + 4 * r) * ld;
        *p = mode == 0 ? cv[y][r] - acc[x][y][r] : acc[x][y][r];
      }
  }
}


// k-loop of the diagonal workgroup (see diag_rhs_tile below), one instantiation per wave W so that everything a
// slot needs is a compile-time constant.  Against the round-2 loop (below, now only the 64-bit-pointer fallback):
//   * both operands come from the SAME panel image and the MFMA's A and B lane maps coincide (lane & 15 -> index,
//     lane >> 4 -> k), so the fragment of block row r IS the fragment of column block r: a k-step loads the column
//     blocks 0 .. max(W, 7 - W) once (plus 2W, 2W + 1 for the right-hand sides where they lie beyond, plus the
//     right-hand-side rows): 7 - 9 ds_reads for 11 MFMAs instead of 14, and no v_cndmask to pick a row fragment;
//   * stage requests are buffer_load ... lds (constant lane offset, rows advance in SGPRs);
//   * reads and requests are dealt out one per MFMA (tests/hip/update_loop_probe.hip).
// Slot s of acc keeps its meaning (s <= W: block row W x column block s; s <= 8: block row 7 - W x column block
// s - W - 1; 9, 10: right-hand-side rows x column blocks 2W, 2W + 1) and every element its k order: same bits.
template <int W>
__device__ __forceinline__ void diag_rhs_accumulate(double* smem, const double* Qp, const double* Tp, int ld,
                                                    int Kdim, d4 (&acc)[11]) {
  constexpr int BKs = 16, TR = 16, STAGE = BKs * kTile + BKs * TR;
  constexpr int RA = W, RB = 7 - W;
  constexpr int CBMAX = RA > RB ? RA : RB;
  constexpr bool XTRA = 2 * W + 1 > CBMAX;             // W = 3: right-hand sides against column blocks 6, 7
  constexpr int NQ = CBMAX + 1 + (XTRA ? 2 : 0);       // panel fragments per k-step; fragment NQ = right-hand-side rows
  constexpr int NR = W < 2 ? 5 : 4;                    // stage requests of this wave (waves 0, 1 also stage T)
  auto frag_of = [](int cb) constexpr { return cb <= CBMAX ? cb : CBMAX + 1 + (cb - 2 * W); };
  auto cb_of_frag = [](int f) constexpr { return f <= CBMAX ? f : 2 * W + (f - CBMAX - 1); };
  const int lane = tid_now() & 63, l15 = lane & 15, l4 = lane >> 4, sw = l4 & 1;

  const __amdgpu_buffer_rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc((void*)Qp, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc((void*)Tp, 0, -1, 0x00020000);
  // panel rows k = W + 4 q, swizzled source (the 16-row block index of row k is XORed with k & 1 = W & 1)
  const unsigned lane_q = (unsigned)(((((lane >> 3) ^ (W & 1)) << 4) + ((lane & 7) << 1)) * 8);
  const unsigned lane_t = (unsigned)((((lane & 7) << 1) + (size_t)(lane >> 3) * ld) * 8);   // 8 columns x 16 rows
  const unsigned stepQ = (unsigned)ld * 32;
  unsigned oQ = (unsigned)W * (unsigned)ld * 8, oT = (unsigned)(8 * W) * (unsigned)ld * 8;
  auto request = [&](int stage, int r) {
    double* St_ = smem + stage * STAGE;
    if (r < 4)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rQ, (__attribute__((address_space(3))) void*)(St_ + (W + 4 * r) * kTile),
                                               16, lane_q, oQ + (unsigned)r * stepQ, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rT, (__attribute__((address_space(3))) void*)(St_ + BKs * kTile + W * 8 * TR),
                                               16, lane_t, oT, 0, 0);
  };
  auto advance = [&]() { oQ += 4 * stepQ; oT += (unsigned)BKs * (unsigned)ld * 8; };

  int off[NQ];
#pragma unroll
  for (int f = 0; f < NQ; ++f) off[f] = l4 * kTile + ((cb_of_frag(f) ^ sw) << 4) + l15;
  const int offT = BKs * kTile + l4 * TR + l15;
  double fA[NQ + 1], fB[NQ + 1];
#define CCGP_SB __builtin_amdgcn_sched_barrier(0)
#define CCGP_DLOAD1(F, STG, KK, I)                                                                \
  do {                                                                                           \
    const double* St_ = smem + (STG) * STAGE;                                                    \
    if ((I) < NQ) F[I] = St_[(KK) * 4 * kTile + off[(I) < NQ ? (I) : 0]];                        \
    else F[NQ] = St_[(KK) * 4 * TR + offT];                                                      \
  } while (0)
#define CCGP_DSTEP(Fc, Fn, STGn, KKn, LD, RQ, STGr)                                               \
  do {                                                                                           \
    _Pragma("unroll") for (int s_ = 0; s_ < 11; ++s_) {                                          \
      const int cb_ = s_ >= 9 ? 2 * W + (s_ - 9) : (s_ <= W ? s_ : s_ - W - 1);                  \
      const double pf_ = s_ >= 9 ? Fc[NQ] : (s_ <= W ? Fc[frag_of(RA)] : Fc[frag_of(RB)]);       \
      acc[s_] = __builtin_amdgcn_mfma_f64_16x16x4f64(Fc[frag_of(cb_)], pf_, acc[s_], 0, 0, 0);   \
      if ((LD) && s_ <= NQ) CCGP_DLOAD1(Fn, STGn, KKn, s_);                                      \
      if ((RQ) && s_ < NR) request(STGr, s_);                                                    \
      CCGP_SB;                                                                                   \
    }                                                                                            \
  } while (0)
#define CCGP_DSTAGE(MORE, REFILL)                                                                 \
  do {                                                                                           \
    const int stg = kt & 1;                                                                      \
    CCGP_DSTEP(fA, fB, stg, 1, true, false, 0);                                                  \
    CCGP_DSTEP(fB, fA, stg, 2, true, false, 0);                                                  \
    CCGP_DSTEP(fA, fB, stg, 3, true, false, 0);                                                  \
    if (MORE) {                                                                                  \
      __syncthreads();                                                                           \
      CCGP_SB;                                                                                   \
    }                                                                                            \
    CCGP_DSTEP(fB, fA, stg ^ 1, 0, MORE, REFILL, stg);                                           \
    if (REFILL) advance();                                                                       \
  } while (0)

  const int nk = Kdim / BKs;
#pragma unroll
  for (int r = 0; r < NR; ++r) request(0, r);
  advance();
  __syncthreads();
  if (nk > 1) {
#pragma unroll
    for (int r = 0; r < NR; ++r) request(1, r);
    advance();
  }
#pragma unroll
  for (int i = 0; i <= NQ; ++i) CCGP_DLOAD1(fA, 0, 0, i);
  CCGP_SB;
  int kt = 0;
  for (; kt < nk - 2; ++kt) CCGP_DSTAGE(true, true);
  if (nk >= 2) { CCGP_DSTAGE(true, false); ++kt; }
  CCGP_DSTAGE(false, false);
#undef CCGP_DSTAGE
#undef CCGP_DSTEP
#undef CCGP_DLOAD1
#undef CCGP_SB
}

// ---- diagonal tile of the update, fused with the right-hand-side rows ---------------------------
// The diagonal tile T_jj = A_jj - L_j L_j' (L_j = the finished panel, rows j of block columns < j)
// needs only its lower triangle: 36 of its 64 16 x 16 sub-tiles, and both GEMM operands are the SAME
// panel.  The right-hand-side rows (y', 1' and 14 zero rows below the matrix) need  b_j' - Z L_j'  against
// that same panel: 8 more sub-tiles.  One workgroup does both: it stages the panel ONCE per k-stage
// (16 KB + 2 KB for the 16 right-hand-side rows instead of 32 KB) and each wave accumulates 11 sub-tiles
// (block rows w and 7 - w of the triangle: 9 sub-tiles whatever w; plus right-hand-side columns 2w, 2w+1)
// against 16 for a full tile.  Before round 2 the diagonal tile ran as a full tile and the right-hand
// sides as a separate "thin" workgroup that occupied a full slot for a whole tile time: at the late
// block columns (few tile rows per matrix) half of the resident workgroups did almost no arithmetic.
__device__ __forceinline__ void diag_rhs_tile(double* smem, const double* Qp, const double* Tp, int ld,
                                              int Kdim, double* C, double* Ct, int wide) {
  constexpr int BKs = 16;
  constexpr int TR = 16;                                // right-hand-side rows staged
  [[maybe_unused]] constexpr int STAGE = BKs * kTile + BKs * TR;         // doubles per stage
  constexpr int NS = 11;                                // sub-tiles per wave: 9 of the triangle + 2 right-hand-side
  const int tid = tid_now(), lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int rA = wave, rB = 7 - wave;

  d4 acc[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) acc[s] = d4{0.0, 0.0, 0.0, 0.0};

  // slot -> column block: slots 0..wave belong to block row rA (cb = s), the rest of 0..8 to block row
  // rB (cb = s - wave - 1); slots 9, 10 are the right-hand-side rows against column blocks 2w, 2w + 1
  int cbs[NS];
#pragma unroll
  for (int s = 0; s < 9; ++s) cbs[s] = s <= wave ? s : s - wave - 1;
  cbs[9] = 2 * wave;
  cbs[10] = 2 * wave + 1;

  // the specialised loop addresses its panel through 32-bit buffer offsets: matrices whose panel spans 4 GiB or
  // more (n >= ~16 000 with the identity rows of an inverse below them) keep the 64-bit-pointer loop -- same bits.
  // That older loop (the `else` branch) is kept for exactly this case and nothing else; CCGP_OPT_WIDE_OFFSETS forces it
  // so that the tests can hold one loop against the other at sizes that fit a test (the OPT_WIDE_OFFSETS test of tests/test_gpu_parity.py).
  if (!wide && fits_buffer_offsets(Kdim, ld)) {
    switch (wave) {
      case 0: diag_rhs_accumulate<0>(smem, Qp, Tp, ld, Kdim, acc); break;
      case 1: diag_rhs_accumulate<1>(smem, Qp, Tp, ld, Kdim, acc); break;
      case 2: diag_rhs_accumulate<2>(smem, Qp, Tp, ld, Kdim, acc); break;
      default: diag_rhs_accumulate<3>(smem, Qp, Tp, ld, Kdim, acc); break;
    }
  } else {
  const int psrc = ((((lane >> 3) ^ (wave & 1)) << 4) + ((lane & 7) << 1));
  const double* pQ = Qp + psrc + (size_t)wave * ld;                       // wave w stages columns k = w + 4q
  const double* pT = Tp + ((lane & 7) << 1) + (size_t)(8 * wave + (lane >> 3)) * ld;   // waves 0, 1: 8 columns each
  const size_t stepQ = (size_t)4 * ld;

  auto issue = [&](int stage) {
    double* Qs_ = smem + stage * STAGE + wave * kTile;
#pragma unroll
    for (int q = 0; q < BKs / 4; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pQ + q * stepQ),
                                       (__attribute__((address_space(3))) void*)(Qs_ + 4 * q * kTile), 16, 0, 0);
    if (wave < 2)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pT,
                                       (__attribute__((address_space(3))) void*)(smem + stage * STAGE + BKs * kTile +
                                                                                 wave * 8 * TR),
                                       16, 0, 0);
    pQ += (BKs / 4) * stepQ;
    pT += (size_t)BKs * ld;
  };

  // element offsets of this lane's fragments inside a stage for k-step 0 (k = l4); k-step kk adds 4 kk rows
  const int sw = l4 & 1;
  const int offA = l4 * kTile + ((rA ^ sw) << 4) + l15;
  const int offB = l4 * kTile + ((rB ^ sw) << 4) + l15;
  const int offT = BKs * kTile + l4 * TR + l15;
  int offQ[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) offQ[s] = l4 * kTile + ((cbs[s] ^ sw) << 4) + l15;

  double pfA[3], qfA[NS], pfB[3], qfB[NS];
#define CCGP_DLOADF(PF, QF, STG, KK)                                                             \
  do {                                                                                          \
    const double* St_ = smem + (STG) * STAGE;                                                   \
    PF[0] = St_[(KK) * 4 * kTile + offA];                                                       \
    PF[1] = St_[(KK) * 4 * kTile + offB];                                                       \
    PF[2] = St_[(KK) * 4 * TR + offT];                                                          \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) QF[s] = St_[(KK) * 4 * kTile + offQ[s]];     \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  } while (0)
#define CCGP_DMFMAS(PF, QF, S0, S1)                                                              \
  do {                                                                                          \
    _Pragma("unroll") for (int s = (S0); s < (S1); ++s) {                                       \
      const double pf = s >= 9 ? PF[2] : (s <= wave ? PF[0] : PF[1]);                           \
      acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(QF[s], pf, acc[s], 0, 0, 0);                \
    }                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  } while (0)
  constexpr int SH = 3;   // MFMAs before the next step's fragment loads are issued

  const int nk = Kdim / BKs;
  issue(0);
  __syncthreads();
  if (nk > 1) issue(1);
  CCGP_DLOADF(pfA, qfA, 0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int stg = kt & 1;
    CCGP_DMFMAS(pfA, qfA, 0, SH);
    CCGP_DLOADF(pfB, qfB, stg, 1);
    CCGP_DMFMAS(pfA, qfA, SH, NS);
    CCGP_DMFMAS(pfB, qfB, 0, SH);
    CCGP_DLOADF(pfA, qfA, stg, 2);
    CCGP_DMFMAS(pfB, qfB, SH, NS);
    CCGP_DMFMAS(pfA, qfA, 0, SH);
    CCGP_DLOADF(pfB, qfB, stg, 3);
    CCGP_DMFMAS(pfA, qfA, SH, NS);
    if (kt + 1 < nk) {
      __syncthreads();
      if (kt + 2 < nk) issue(stg);
    }
    CCGP_DMFMAS(pfB, qfB, 0, SH);
    if (kt + 1 < nk) CCGP_DLOADF(pfA, qfA, stg ^ 1, 0);
    CCGP_DMFMAS(pfB, qfB, SH, NS);
  }
#undef CCGP_DLOADF
#undef CCGP_DMFMAS
  }

  // C -= acc: all loads of a group before its stores (see gemm_tile)
#pragma unroll
  for (int g0 = 0; g0 < NS; g0 += 4) {
    double cv[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int s = g0 + u;
      if (s >= NS) break;
      const int rb = s >= 9 ? 0 : (s <= wave ? rA : rB);
      double* base = s >= 9 ? Ct : C;
#pragma unroll
      for (int r = 0; r < 4; ++r) cv[u][r] = base[rb * 16 + l15 + (size_t)(cbs[s] * 16 + l4 + 4 * r) * ld];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int s = g0 + u;
      if (s >= NS) break;
      const int rb = s >= 9 ? 0 : (s <= wave ? rA : rB);
      double* base = s >= 9 ? Ct : C;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        base[rb * 16 + l15 + (size_t)(cbs[s] * 16 + l4 + 4 * r) * ld] = cv[u][r] - acc[s][r];
    }
  }
}

template <int S>
constexpr size_t gemm_lds_bytes() {
  return sizeof(double) * 2 * TileGeom<S, false>::BK * (kTile + kTile / S);   // two unpadded stages
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
  double ptol;          // pivot_tolerance(mean_mode, n), ccgp_internal.h
};

// Register-resident: the 256 threads form a 16 x 16 grid (ty = row class, tx = column class)
// and own the block 2-D cyclically -- thread (ty, tx) holds entries (ty + 16a, tx + 16b).  The
// 128 x 128 block is factorised as L' D L'^T with the identity appended as 128 extra rows, so
// the same rank-1 sweep that eliminates column k also produces L'^-1 (forward substitution of
// e_t): per column one barrier, one 256-word column broadcast through LDS, and <= 44 FMAs per
// thread on registers.  Then L = L' D^1/2 and W = L^-1 = D^-1/2 L'^-1 are written out.
// lds: 2 * 256 + 128 + 4 doubles of scratch (kDiagLdsDoubles)
constexpr int kDiagLdsDoubles = 2 * 256 + kTile + 4;
__device__ __forceinline__ void diag_factor(const DiagArgs& g, int b, double* lds) {
  double (*colbuf)[256] = reinterpret_cast<double (*)[256]>(lds);
  double* dvec = lds + 2 * 256;
  double* red = dvec + kTile;
  const int tid = tid_now(), ty = tid & 15, tx = tid >> 4;
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
      if (!(piv > g.ptol)) { bad = k + 1; break; }   // uniform: every thread reads the same word
      // 1 / pivot by v_rcp_f64 + two Newton steps (full precision, a fifth of a division's instructions)
      double rinv = __builtin_amdgcn_rcp(piv);
      rinv = fma(fma(-piv, rinv, 1.0), rinv, rinv);
      rinv = fma(fma(-piv, rinv, 1.0), rinv, rinv);
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

__global__ __launch_bounds__(256) void diag_kernel(DiagArgs g) {
  __shared__ double lds[kDiagLdsDoubles];
  diag_factor(g, blockIdx.x, lds);
}

// Workgroups of one launch.
// update (MODE 0): first ONE workgroup per matrix for the diagonal tile + right-hand sides (diag_rhs_tile; it goes
// on to factorise the block, so it is dispatched first), then the tile rows below the diagonal and the extra tile
// rows (prediction: r(x_t)'; inverse / gradient: identity), matrices in groups of eight, each tile as S column
// strips.  In the S = 1 kernel the LAST `tail` tiles (in that order) may run as two half-width ring-pipelined
// strips each: the last, partial step of 256 workgroups then consists of strips only and takes about half a
// tile time (update_tail()).
// trsm (MODE 1): per matrix tile rows j+1 .. nt-1, the thin right-hand-side tile row nt, then the extra rows.
// rows_only (kept factor): the extra rows alone.
// Block index % 8 = matrix % 8 everywhere: blocks are dealt round-robin over the 8 XCDs, so all tiles of one matrix
// share the column panel in that XCD's L2.
__host__ __device__ inline int gemm_units_per_matrix(int mode, int nt, int j, int ne, int S, int rows_only = 0) {
  if (rows_only) return mode == 0 ? ne * S : ne;
  return mode == 0 ? 1 + ((nt - 1 - j) + ne) * S : (nt - j) + ne;
}

// One unit of block column j for matrix b: tile row i (i == nt: trsm's thin right-hand-side row; i > nt: extra rows),
// column strip `strip` of S (or of g.tail_s ring strips).  Shared by the launch-per-phase kernels (gemm_dispatch maps
// blockIdx to a unit) and the dataflow scheduler (chol_sched_kernel takes units from a queue).
template <int MODE, int S>
__device__ __forceinline__ void gemm_unit(const GemmArgs& g, double* smem, int j, int b, int i, int strip, bool ring) {
  const int ld = g.ld;
  double* Ab = g.A + (size_t)b * g.a_stride;
  const bool thin = i == g.nt;
  const int c0 = strip * (ring ? kTile / g.tail_s : kTile / S);   // first column of this strip inside the tile

  // identity rows: block row te of Z = L^-T starts at block column te, so tiles left of it are
  // zero (skipped) and the update's k-sum starts at te
  int k0 = 0;
  if (g.extra_lower && i > g.nt) {
    const int te = i - g.nt - 1;
    if (j < te || (MODE == 0 && j == te)) return;
    k0 = te * kTile;
  }
  // tile row i: rows i*128.. of the matrix array, or -- extra rows kept in their own buffer -- of E
  double* rowbase = Ab + (size_t)i * kTile;
  int rld = ld;
  if (g.E && i > g.nt) {
    rowbase = g.E + (size_t)b * g.e_stride + (size_t)(i - g.nt - 1) * kTile;
    rld = g.lde;
  }
  const double* P;
  const double* Q;
  int ldP, ldQ, Kdim;
  if (MODE == 0) {
    P = rowbase + (size_t)k0 * rld;
    Q = Ab + (size_t)j * kTile + c0 + (size_t)k0 * ld;
    ldP = rld;
    ldQ = ld;
    Kdim = j * kTile - k0;
  } else {
    P = rowbase + (size_t)j * kTile * rld;
    Q = g.invd + (size_t)b * g.invd_stride + (size_t)j * kTile * kTile + c0;
    ldP = rld;
    ldQ = kTile;
    Kdim = kTile;
  }
  double* C = rowbase + ((size_t)j * kTile + c0) * rld;
  constexpr bool TRI = MODE == 1 && S == 1;   // trsm: Q is the lower-triangular inverse block
  if constexpr (MODE == 1) {
    if (thin) { gemm_tile<S, true, TRI>(smem, P, ldP, Q, ldQ, Kdim, C, rld, MODE); return; }
  }
  if constexpr (MODE == 0 && S == 1) {
    if (ring) {
      if (g.tail_s == 4) gemm_tile<4, false, false, true>(smem, P, ldP, Q, ldQ, Kdim, C, rld, MODE);
      else gemm_tile<2, false, false, true>(smem, P, ldP, Q, ldQ, Kdim, C, rld, MODE);
      return;
    }
    if (!g.wide && fits_buffer_offsets(Kdim, ldP > ldQ ? ldP : ldQ)) {   // else: 64-bit-pointer loop, same bits
      update_tile_il(smem, P, ldP, Q, ldQ, Kdim, C, rld);
      return;
    }
  }
  gemm_tile<S, false, TRI>(smem, P, ldP, Q, ldQ, Kdim, C, rld, MODE);
}

// The diagonal workgroup of block column j >= 1 for matrix b: T_jj and the right-hand-side rows minus the finished panel,
// then (fuse_diag) the factorisation and inverse of the block.
__device__ __forceinline__ void diag_unit(const GemmArgs& g, double* smem, int j, int b) {
  const int ld = g.ld;
  double* Ab = g.A + (size_t)b * g.a_stride;
  diag_rhs_tile(smem, Ab + (size_t)j * kTile, Ab + g.npad, ld, j * kTile,
                Ab + (size_t)j * kTile + (size_t)j * kTile * ld, Ab + g.npad + (size_t)j * kTile * ld, g.wide);
  if (g.fuse_diag) {
    // T_jj went to memory through this CU's L1; the waves of the workgroup read it back in diag_factor's
    // thread layout.  All waves of a workgroup share that L1, so workgroup scope is enough: __syncthreads is
    // release(workgroup) + s_barrier + acquire(workgroup) -- no L2 write-back, no cache invalidate
    __syncthreads();
    DiagArgs dg{g.A, g.a_stride, g.npad, g.invd, g.invd_stride, g.logdet_part, g.status, j, g.nt, g.nb, g.n, g.ld, g.ptol};
    diag_factor(dg, b, smem);   // the staging LDS is free: the K loop ended on a barrier
  }
}

template <int MODE, int S>
__device__ __forceinline__ void gemm_dispatch(const GemmArgs& g, double* smem) {
  const int L = blockIdx.x;
  int b, i, strip = 0;
  bool ring = false;
  if (MODE == 0 && !g.rows_only) {
    const int nb8 = (g.nb + 7) & ~7;
    if (L < nb8) {
      if (L < g.nb) diag_unit(g, smem, g.j, L);
      return;
    }
    const int nfull = (g.nt - 1 - g.j) + g.ne;          // tiles per matrix
    const int Lt = L - nb8;
    int u1;                                             // tile index in (group, tile, matrix-in-group) order
    if (S == 1) {
      if (Lt >= g.n_s1) {                               // tail: q = 8 tail_s a + 8 strip + m  ->  tile n_s1 + 8 a + m
        const int q = Lt - g.n_s1;
        u1 = g.n_s1 + ((q / (8 * g.tail_s)) << 3) + (q & 7);
        strip = (q >> 3) % g.tail_s;
        ring = true;
      } else {
        u1 = Lt;
      }
    } else {                                            // every tile as S strips: q = 8 S a + 8 strip + m
      u1 = ((Lt / (8 * S)) << 3) + (Lt & 7);
      strip = (Lt >> 3) % S;
    }
    const int per_grp = 8 * nfull;
    const int grp = u1 / per_grp, r = u1 % per_grp;
    b = grp * 8 + (r & 7);
    if (b >= g.nb) return;
    i = g.j + 1 + (r >> 3);
    if (i >= g.nt) i += 1;   // the right-hand-side tile row nt went with the diagonal tile
  } else {
    const int per_grp = 8 * gemm_units_per_matrix(MODE, g.nt, g.j, g.ne, S, g.rows_only);
    const int grp = L / per_grp, r = L % per_grp;
    b = grp * 8 + (r & 7);
    if (b >= g.nb) return;
    const int u = r >> 3;   // unit index inside the matrix
    if (g.rows_only) {
      strip = MODE == 0 ? u % S : 0;
      i = g.nt + 1 + (MODE == 0 ? u / S : u);
    } else {
      i = g.j + 1 + u;
    }
  }
  gemm_unit<MODE, S>(g, smem, g.j, b, i, strip, ring);
}

// distinct kernel symbols per phase (rocprof attributes time per symbol) and per strip count
#define CCGP_DEFINE_GEMM(NAME, MODE, S, WPS)                                            \
  __global__ __launch_bounds__(256, WPS) void NAME(GemmArgs g) {                        \
    extern __shared__ __attribute__((aligned(16))) double smem[];                       \
    gemm_dispatch<MODE, S>(g, smem);                                                     \
  }
CCGP_DEFINE_GEMM(chol_update_kernel, 0, 1, 2)
CCGP_DEFINE_GEMM(chol_update_s2_kernel, 0, 2, 2)
// trsm exists at S = 1 only: it is in place (reads the whole tile row, writes its own columns), so column
// strips of one tile would race
CCGP_DEFINE_GEMM(chol_trsm_kernel, 1, 1, 2)
#undef CCGP_DEFINE_GEMM

#include "blocked_sched.inc"   // chol_sched_kernel, sched_init_kernel, sched_check_kernel (round 5)

// Strip count of an update launch: always 1 since round 2.  One workgroup per SIMD-set saturates a CU's four MFMA
// pipes, so the time of a launch is a step function of its workgroup count in units of 256 (16.3 us per 128-deep
// block of K and per step; the second resident workgroup of a CU only fills bubbles); with the diagonal tile as ONE
// long workgroup (it also factorises the block) whole launches of half-width strips no longer win anywhere
// (profiles/r02_update_schedule.md sections 3 and 6).  The S = 2 kernel remains for the rows-only sweeps of
// ccgp_predict_from_factorset (few, long rows).

// Tiles of an update launch that should run as strips: the launch is W1 = nb8 (1 + tiles) workgroups at S = 1 and
// takes ceil(W1 / 256) steps (one workgroup per CU saturates its MFMA pipes).  If the last step holds `rem` <= 128
// TILES (the diagonal workgroups are dispatched first and are never in it unless the whole launch is one step),
// running those tiles as 2 rem ring-pipelined half-width strips on otherwise idle CUs cuts that step to about half.
// Round 4: quarter-width strips where they fill the step -- four per tail tile if that leaves at most two workgroups per CU
// in the last step (a quarter step of tiles: 256 strips on 256 CUs; half a step: 512, two per CU); `quarters` = 0 keeps the
// round-3 policy (CCGP_OPT_TAIL_STRIPS = 2).
static int update_tail(int nb8, int tiles, int quarters, int* ts) {
  const long w1 = (long)nb8 * (1 + tiles), all_tiles = (long)nb8 * tiles;
  const long rem = w1 % 256;
  *ts = 2;
  if (rem == 0 || all_tiles == 0) return 0;
  long tail = rem < all_tiles ? rem : all_tiles;
  if (tail != rem && w1 > 256) return 0;             // whole tiles left in the last step: nothing gained
  // (only behind at least one full step: a launch that is a single partial step is bound by its diagonal workgroups, and
  // quarter strips there cost 0.13 ms at block column 30 of the 64-matrix slice, 3 % of a 16-matrix batch)
  if (quarters && w1 > 256 && (rem - tail) + 4 * tail <= 512) { *ts = 4; return (int)tail; }
  if (rem + tail > 256) return 0;                    // the strips would not fit the same step
  return (int)tail;                                  // a multiple of 8 (nb8 is)
}

static void launch_gemm(hipStream_t s, GemmArgs g, int mode, int S) {
  const int nb8 = round_up(g.nb, 8);
  int units = nb8 * gemm_units_per_matrix(mode, g.nt, g.j, g.ne, S, g.rows_only);
  if (mode == 0 && !g.rows_only) {
    const int tiles = (g.nt - 1 - g.j) + g.ne;
    int ts = 2;
    const int tail = (S == 1 && g.tail_strips) ? update_tail(nb8, tiles, g.tail_strips == 1, &ts) : 0;
    g.n_s1 = nb8 * tiles - tail;
    g.tail_s = ts;
    units = nb8 + (S == 1 ? g.n_s1 + ts * tail : nb8 * tiles * S);
  }
  const dim3 grid(units), block(256);
  if (mode == 0) {
    if (S == 1) hipLaunchKernelGGL(chol_update_kernel, grid, block, gemm_lds_bytes<1>(), s, g);
    else hipLaunchKernelGGL(chol_update_s2_kernel, grid, block, gemm_lds_bytes<2>(), s, g);
  } else {
    hipLaunchKernelGGL(chol_trsm_kernel, grid, block, gemm_lds_bytes<1>(), s, g);
  }
}

// ---- right-hand-side rows and the final reductions ---------------------------------------------
struct RhsArgs {
  double* A;
  size_t a_stride;
  int npad, n;
  const double* y;
  int ld;
  int identity;   // extra rows = identity (inverse / gradient) instead of zeros (prediction)
};

// rows npad..npad+15 of every matrix: y' (zero beyond n), 1' (zero beyond n), 14 zero rows.  Only these 16 of the
// tile row's 128 rows are ever read (the diagonal workgroup and the thin panel-solve tile stage 16 right-hand-side
// rows; finish / alpha / predict read rows 0 and 1): writing all 128 was 2.1 GB and 0.45 ms per 512 matrices.
// One workgroup = 16 columns x 16 rows.
__global__ __launch_bounds__(256) void rhs_rows_kernel(RhsArgs g) {
  const int ld = g.ld;
  double* Ab = g.A + (size_t)blockIdx.y * g.a_stride;
  const int c = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int r = threadIdx.x & 15;
  double v = 0.0;
  if (c < g.n) v = r == 0 ? g.y[c] : (r == 1 ? 1.0 : 0.0);
  Ab[g.npad + r + (size_t)c * ld] = v;
  // extra tile rows start as zeros (the cross-correlation kernel then fills rows < m, columns < n).  Identity rows: row t
  // stays zero left of column t, so the sweep, rinv_tile_kernel and alpha_kernel never touch a tile whose row block lies
  // below its column block -- only the tiles on and above that diagonal are written (half of 134 MB per matrix at n = 4096)
  const int e_end = g.identity ? min(ld, g.npad + kTile + (c / kTile + 1) * kTile) : ld;
  for (int e = g.npad + kTile + r; e < e_end; e += 16)
    Ab[e + (size_t)c * ld] = (g.identity && e - (g.npad + kTile) == c) ? 1.0 : 0.0;
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
  double* logdet_out;   // log det of the factorised matrix (mode 0: the normalised R), indexed like loglik; or nullptr
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
    if (g.logdet_out) g.logdet_out[gb] = (g.status && g.status[gb] != 0) ? kNaN : logdet;
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
  const double* E;   // rows L^-1 r(x_t) when they are kept outside the matrix array (else nullptr)
  size_t e_stride;
  int lde;
  const double* s11;
  const double* beta;
  const int* status;
  int b0, S;
  double sigma2;
  double* mean;
  double* var;
};

// A workgroup owns 64 test sites (lane = site: a column of the rows L^-1 r(x_t) is 512 contiguous bytes per wave), its 16
// waves take the columns c = w, w + 16, ...; three running sums per lane, combined in LDS in fixed order.  (Rounds 1 - 3:
// one thread per site walking all n columns alone -- 1.2 ms for 16 draws x 128 sites at n = 4096, a tenth of the call.)
constexpr int kPredFinishWaves = 16;
__global__ __launch_bounds__(64 * kPredFinishWaves) void predict_finish_kernel(PredFinishArgs g) {
  __shared__ double part[3][kPredFinishWaves][64];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = blockIdx.x * 64 + lane;
  const double* Ab = g.A + (size_t)b * g.a_stride;
  const double* zrow = Ab + g.npad;
  const int tc = t < g.m ? t : g.m - 1;   // lanes beyond m read a valid row and write nothing
  const double* wrow = g.E ? g.E + (size_t)b * g.e_stride + tc : Ab + g.npad + kTile + tc;
  const int ldw = g.E ? g.lde : g.ld;
  double ww = 0.0, z1w = 0.0, zyw = 0.0;
#pragma unroll 4
  for (int c = wave; c < g.n; c += kPredFinishWaves) {
    const double w = wrow[(size_t)c * ldw];
    ww = fma(w, w, ww);
    z1w = fma(zrow[1 + (size_t)c * g.ld], w, z1w);
    zyw = fma(zrow[(size_t)c * g.ld], w, zyw);
  }
  part[0][wave][lane] = ww;
  part[1][wave][lane] = z1w;
  part[2][wave][lane] = zyw;
  __syncthreads();
  if (tid >= 64 || t >= g.m) return;
  ww = z1w = zyw = 0.0;
  for (int w = 0; w < kPredFinishWaves; ++w) {
    ww += part[0][w][lane];
    z1w += part[1][w][lane];
    zyw += part[2][w][lane];
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

// ---- explicit inverse and gradient from the identity rows ----------------------------------------
// After a sweep with the identity as extra rows, row t of the extra block is Z[t][c] = (L^-1)[c][t]
// (zero for c < t), i.e. Z = L^-T, and  R^-1 = Z Z'.
struct AlphaArgs {
  const double* A;
  size_t a_stride;
  int npad, ld, n;
  const double* beta;   // chunk-local (fin + nb)
  double* alpha;        // nb x npad:  R^-1 (y - beta 1) = Z (z_y - beta z_1)
};

// A workgroup owns 64 rows (lane = row: a column of Z is read as 512 contiguous bytes per wave), its 16 waves take the
// columns c = a0 + w, a0 + w + 16, ...; fixed summation order.  (Rounds 2 - 3: one thread per row walking its up to n columns
// alone -- 3.2 ms for 16 draws at n = 4096, half as long as forming R^-1.)
constexpr int kAlphaWaves = 16;
__global__ __launch_bounds__(64 * kAlphaWaves) void alpha_kernel(AlphaArgs g) {
  __shared__ double part[kAlphaWaves][64];
  const int b = blockIdx.y, a0 = blockIdx.x * 64, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int a = a0 + lane;
  const double* Ab = g.A + (size_t)b * g.a_stride;
  const double* zrow = Ab + g.npad;
  const double* Zrow = Ab + g.npad + kTile + a;
  const double beta = g.beta[b];
  double s = 0.0;
  if (a < g.n) {
#pragma unroll 4
    for (int c = a0 + wave; c < g.n; c += kAlphaWaves) {
      const double z = Zrow[(size_t)c * g.ld];
      const double v = zrow[(size_t)c * g.ld] - beta * zrow[1 + (size_t)c * g.ld];
      if (c >= a) s = fma(z, v, s);
    }
  }
  part[wave][lane] = s;
  __syncthreads();
  if (tid < 64) {
    double t = 0.0;
    for (int w = 0; w < kAlphaWaves; ++w) t += part[w][tid];
    g.alpha[(size_t)b * g.npad + a0 + tid] = t;
  }
}

struct RinvArgs {
  const double* A;
  size_t a_stride;
  int npad, ld, nt, n, nb;
  int wide;              // CCGP_OPT_WIDE_OFFSETS: the 64-bit-pointer loop everywhere
  // inverse
  double* Rinv;          // nb matrices of n x n
  // gradient
  const double* X;
  int d, K;
  const double* params;
  int ldp, draw0;
  double sigma2;
  const double* alpha;   // nb x npad
  double* gpart;         // nb x ntiles x P
};

// One 128 x 128 tile (ta >= tb) of R^-1 = Z Z' (k from block ta on, where both row blocks are non-zero), kept in
// registers and either written out symmetrically (GRAD = false: solve(R), HX:454) or turned into the tile of
//   M = (alpha alpha' - R^-1 / c) / 2,   c = sigma2 sum w^2,   alpha = R^-1 (y - beta 1) / c
// and stored over the (no longer needed) tile (ta, tb) of L, zero outside the n x n matrix (GRAD = true); the
// contraction with the kernel derivatives is grad_contract_kernel's.
// Round 4: the product runs on the update's round-3 loop (tile_accumulate_il: buffer-offset stage requests, b128 fragment
// reads; the round-2 loop remains for panels of 4 GiB and more, same bits) with TWO workgroups per CU.  Rounds 2 - 3
// contracted in this kernel's epilogue: the tile of M .* R_q in a second set of 128 registers beside the accumulators
// (512 VGPRs: one workgroup per CU, the round-2 loop alone on its CU -- 30 TFLOP/s), and every attempt to contract without
// that second set under 256 registers ended in kilobytes of scratch (14.6 -> 25.6 ms per 16 draws at n = 4096).  A round
// trip of M through HBM costs 2 x 67 MB per draw -- 0.03 ms at 5 TB/s -- and leaves each kernel with one job.
template <bool IL>
struct RinvMap {
  int row0, col0, l15, l4;
  __device__ RinvMap() {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    row0 = (wave >> 1) * 64; col0 = (wave & 1) * 64; l15 = lane & 15; l4 = lane >> 4;
  }
  __device__ __forceinline__ int row(int y) const { return IL ? row0 + 4 * l15 + y : row0 + 16 * y + l15; }
  __device__ __forceinline__ int col(int x, int r) const { return IL ? col0 + 16 * r + 4 * l4 + x : col0 + 16 * x + l4 + 4 * r; }
};

template <bool GRAD, bool IL>
__device__ __forceinline__ void rinv_tile_finish(const RinvArgs& g, d4 (&acc)[4][4], int b, int ta, int tb) {
  // acc[x][y][r] = Rinv[row = ta*128 + map.row(y)][col = tb*128 + map.col(x, r)]
  const RinvMap<IL> map;
  const int n = g.n;
  if constexpr (!GRAD) {
    double* out = g.Rinv + (size_t)b * n * n;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ga = ta * kTile + map.row(y), gb = tb * kTile + map.col(x, r);
          if (ga < n && gb < n) {
            out[ga + (size_t)gb * n] = acc[x][y][r];
            out[gb + (size_t)ga * n] = acc[x][y][r];
          }
        }
  } else {
    const int gdraw = g.draw0 + b;
    double sw = 0.0;
    for (int q = 0; q < g.K; ++q) { const double w = g.params[gdraw + (size_t)q * g.ldp]; sw += w * w; }
    const double cs = g.sigma2 * sw;
    const double* al = g.alpha + (size_t)b * g.npad;
    double* Mt = const_cast<double*>(g.A) + (size_t)b * g.a_stride + (size_t)ta * kTile + (size_t)tb * kTile * g.ld;
    double ala[4];
#pragma unroll
    for (int y = 0; y < 4; ++y) ala[y] = al[ta * kTile + map.row(y)];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lc = map.col(x, r), gb = tb * kTile + lc;
        const double alb = al[gb];
        d4 o;
#pragma unroll
        for (int y = 0; y < 4; ++y) {
          const int ga = ta * kTile + map.row(y);
          const double m = 0.5 * (ala[y] * alb / (cs * cs) - acc[x][y][r] / cs);
          o[y] = (ga < n && gb < n) ? m : 0.0;
        }
        if constexpr (IL) {
          *(d4*)(Mt + map.row(0) + (size_t)lc * g.ld) = o;     // a lane's four rows are contiguous
        } else {
#pragma unroll
          for (int y = 0; y < 4; ++y) Mt[map.row(y) + (size_t)lc * g.ld] = o[y];
        }
      }
  }
}

template <bool GRAD>
__global__ __launch_bounds__(256, 2) void rinv_tile_kernel(RinvArgs g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int ntiles = g.nt * (g.nt + 1) / 2;
  const int L = blockIdx.x;
  const int per_grp = 8 * ntiles;
  const int grp = L / per_grp, rr = L % per_grp;
  const int b = grp * 8 + (rr & 7);
  const int t = rr >> 3;
  if (b >= g.nb) return;
  int ta = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((ta + 1) * (ta + 2) / 2 <= t) ++ta;
  while (ta * (ta + 1) / 2 > t) --ta;
  const int tb = t - ta * (ta + 1) / 2;

  const double* Zb = g.A + (size_t)b * g.a_stride + g.npad + kTile;
  const double* P = Zb + (size_t)ta * kTile + (size_t)ta * kTile * g.ld;
  const double* Q = Zb + (size_t)tb * kTile + (size_t)ta * kTile * g.ld;
  const int Kdim = g.npad - ta * kTile;
  d4 acc[4][4];
  if (!g.wide && fits_buffer_offsets(Kdim, g.ld)) {   // uniform
    tile_accumulate_il(smem, P, g.ld, Q, g.ld, Kdim, acc);
    rinv_tile_finish<GRAD, true>(g, acc, b, ta, tb);
  } else {                                            // 64-bit-pointer loop, same bits
    gemm_accumulate<1, false>(smem, P, g.ld, Q, g.ld, Kdim, acc);
    rinv_tile_finish<GRAD, false>(g, acc, b, ta, tb);
  }
}

// ---- gradient: M contracted with the kernel derivatives ------------------------------------------------
//   d loglik / d w_q      =  2 sigma2 w_q     sum_ab M_ab R_q,ab
//   d loglik / d theta_qk = -  sigma2 w_q^2   sum_ab M_ab (x_ak - x_bk)^2 R_q,ab
// over the lower 64 x 64 tiles of M (rinv_tile_kernel<true> left them where L was; zero beyond n); a tile off the diagonal
// counts twice, a diagonal tile is summed whole (M and R_q are symmetric).  Shaped like cov_kernel, whose arithmetic it
// repeats (R_q is regenerated from X, never stored): lane = row, a wave owns 16 columns, the column coordinates are
// broadcast LDS reads, one component per pass with 1 + KG running sums per lane (d <= KG = 4, 6, 8; theta_qk = 0 beyond d
// makes the padded dimensions vanish from the dot product).  Vector-issue bound like cov_kernel: ~38 instructions per
// entry and component.  KG = 0: any d, run-time loops over the dimensions in groups of eight (the exp is recomputed per
// group).  Partial sums per tile -> gpart[(matrix, tile)][P], summed by blocked_grad_reduce_kernel.
struct GradContractArgs {
  const double* A;       // M tiles (lower), column-major with leading dimension ld
  size_t a_stride;
  int ld, n, d, K;
  const double* X;       // n x d
  const double* params;
  int ldp, draw0;
  double* gpart;         // nb x ntiles64 x P
  int ntiles64;
};

size_t grad_contract_lds(int d, int K) {
  return sizeof(double) * ((size_t)kExpTableDoubles + (size_t)(2 * d + 2 * K) * 64 + (size_t)K * d + 4 * (size_t)(K + K * d));
}

template <int KG>
__global__ __launch_bounds__(256, (KG == 4 || KG == 6) ? 4 : 3) void grad_contract_kernel(GradContractArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int G = KG > 0 ? KG : 8;
  const int d = a.d, K = a.K, Pn = K + K * d, n = a.n;
  double* etab = smem;
  double* xa = etab + kExpTableDoubles;   // [d][64]
  double* xb = xa + d * 64;               // [d][64]
  double* ua = xb + d * 64;               // [K][64]
  double* ub = ua + K * 64;               // [K][64]
  double* th = ub + K * 64;               // [K][d]
  double* part = th + K * d;              // [4][Pn]
  const int t = blockIdx.x, b = blockIdx.z, gdraw = a.draw0 + b;
  int tr = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((tr + 1) * (tr + 2) / 2 <= t) ++tr;
  while (tr * (tr + 1) / 2 > t) --tr;
  const int tc = t - tr * (tr + 1) / 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i0 = tr * 64, j0 = tc * 64, jl0 = wave * 16;

  // this lane's 16 entries of M first: the only HBM reads of the kernel, in flight during the prologue
  const double* Mt = a.A + (size_t)b * a.a_stride + (size_t)(i0 + lane) + (size_t)(j0 + jl0) * a.ld;
  double m[16];
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) m[jj] = Mt[(size_t)jj * a.ld];

  exp_table_load(etab, tid, 256);
  for (int e = tid; e < K * d; e += 256) th[e] = a.params[gdraw + (size_t)(K + e) * a.ldp];
  for (int e = tid; e < d * 64; e += 256) {
    const int k = e >> 6, r = e & 63;
    xa[e] = i0 + r < n ? a.X[i0 + r + (size_t)k * n] : 0.0;
    xb[e] = j0 + r < n ? a.X[j0 + r + (size_t)k * n] : 0.0;
  }
  __syncthreads();
  for (int e = tid; e < K * 64; e += 256) {
    const int q = e >> 6, r = e & 63;
    double sa = 0.0, sb = 0.0;
    for (int k = 0; k < d; ++k) {
      const double tq = th[q * d + k], va = xa[k * 64 + r], vb = xb[k * 64 + r];
      sa = fma(va * va, tq, sa);
      sb = fma(vb * vb, tq, sb);
    }
    ua[e] = sa;
    ub[e] = sb;
  }
  __syncthreads();

  for (int q = 0; q < K; ++q) {
    const double ur = ua[q * 64 + lane];
    for (int k0 = 0; k0 < (KG > 0 ? 1 : d); k0 += G) {
      double xav[G], xr[G], hs[G], gs = 0.0;
#pragma unroll
      for (int kk = 0; kk < G; ++kk) {
        const int k = k0 + kk;
        xav[kk] = xa[(k < d ? k : d - 1) * 64 + lane];
        // the squared differences do not depend on the component: unpinned, those of all 16 columns are computed before
        // the loop over the components and kept (16 KG doubles per lane: scratch)
        asm volatile("" : "+v"(xav[kk]));
        xr[kk] = k < d ? xav[kk] * th[q * d + k] : 0.0;
        hs[kk] = 0.0;
      }
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const int c = jl0 + jj;
        double sdot = 0.0, dfsq[G];
#pragma unroll
        for (int kk = 0; kk < G; ++kk) {
          const int k = k0 + kk;
          const double xbv = xb[(k < d ? k : d - 1) * 64 + c];
          const double df = xav[kk] - xbv;
          dfsq[kk] = df * df;
          if (KG > 0) sdot = fma(xr[kk], xbv, sdot);
        }
        if (KG == 0)
          for (int k = 0; k < d; ++k) sdot = fma(xa[k * 64 + lane] * th[q * d + k], xb[k * 64 + c], sdot);
        const double dist = fma(-2.0, sdot, ur + ub[q * 64 + c]);
        const double v = m[jj] * exp_cov(dist, etab);
        gs += v;
#pragma unroll
        for (int kk = 0; kk < G; ++kk) {
          hs[kk] = fma(v, dfsq[kk], hs[kk]);
          // pinned: the running sums are only read after the loop, so left alone their FMAs are sunk below all 16 exp
          // chains -- with the 16 values and 16 KG column coordinates they need kept in scratch until then
          asm volatile("" : "+v"(hs[kk]));
        }
      }
      if (k0 == 0) {
        for (int off = 32; off > 0; off >>= 1) gs += __shfl_down(gs, off, 64);
        if (lane == 0) part[wave * Pn + q] = gs;
      }
#pragma unroll
      for (int kk = 0; kk < G; ++kk) {
        double v = hs[kk];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0 && k0 + kk < d) part[wave * Pn + K + q * d + k0 + kk] = v;
      }
    }
  }
  __syncthreads();
  if (tid < Pn) {
    const double sym = tr == tc ? 1.0 : 2.0;
    a.gpart[((size_t)b * a.ntiles64 + t) * Pn + tid] =
        sym * ((part[tid] + part[Pn + tid]) + (part[2 * Pn + tid] + part[3 * Pn + tid]));
  }
}

struct GradReduceArgs {
  const double* gpart;
  int ntiles, P, K, d, nb, b0, Btot;
  const double* params;
  int ldp;
  double sigma2;
  const int* status;
  double* grad;
};

// one workgroup per (matrix, parameter): 256 strided partial sums, then the block's fixed-order tree (round 4; before, ONE
// thread walked a parameter's 528 -- now 2080 -- partials: 0.9 ms for 16 draws at n = 4096, as long as the contraction itself)
__global__ __launch_bounds__(256) void blocked_grad_reduce_kernel(GradReduceArgs g) {
  __shared__ double red[4];
  const int b = blockIdx.x / g.P, q = blockIdx.x % g.P, gb = g.b0 + b, tid = threadIdx.x;
  double s = 0.0;
  for (int t = tid; t < g.ntiles; t += 256) s += g.gpart[((size_t)b * g.ntiles + t) * g.P + q];
  s = block_sum(s, red, tid);
  if (tid) return;
  const int c = q < g.K ? q : (q - g.K) / g.d;
  const double wc = g.params[gb + (size_t)c * g.ldp];
  double v = q < g.K ? 2.0 * g.sigma2 * wc * s : -g.sigma2 * wc * wc * s;
  if (g.status && g.status[gb] != 0) v = __longlong_as_double(0x7ff8000000000000LL);
  g.grad[gb + (size_t)q * g.Btot] = v;
}

}  // namespace

size_t blocked_ws_bytes(int npad, int nb, int ne) {
  const int nt = npad / kTile;
  size_t dbl = (size_t)nb * (npad + kTile * (1 + ne)) * npad + (size_t)nb * nt * kTile * kTile +
               (size_t)nb * (nt + 2) + 64 + (size_t)npad * kMaxD + 16 +
               (size_t)nb * kMaxK * npad + 16;   // upad
  return dbl * sizeof(double) + sched_ws_bytes(nt, nb, ne);
}

// scratch of the dataflow scheduler: task slots (8 B per task), counters, queue control block, per-CU registration
size_t sched_ws_bytes(int nt, int nb, int ne) {
  const sched::Shape s{nt, ne, 0};   // identity rows (lower) only take tasks away: an upper bound
  return 8 * (size_t)nb * (size_t)sched::tasks_per_matrix(s) + 4 * ((size_t)nb * sched::counters_per_matrix(s) + kSchedCtrlInts + 8 * 128) + 64 +
         8 * (size_t)kSchedProfWgs * kSchedProfWords;
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
  w.xpad = w.fin + (((size_t)2 * nb + 15) & ~(size_t)15);   // 128-byte aligned: whole s_load_dwordx16 lines
  w.upad = w.xpad + (size_t)npad * kMaxD;
  w.sched = w.upad + (size_t)nb * kMaxK * npad + 16;
  return w;
}

// One matrix group's whole sweep, enqueued on stream s.  w is already offset to the group.
// One chunk's sweep (all matrices advance together), split into its phases.
struct GroupRun {
  ccgp_handle* h;
  hipStream_t s;
  const double* X;
  int n, d;
  const double* y;
  DrawView dv;
  int b0, nb, npad, nt;
  double sigma2;
  int mean_mode;
  double tau2;
  BlockedWs w;
  double* loglik;
  double* beta;
  int* status;
  const BlockedJob* job;
  GemmArgs g{};
  DiagArgs dg{};
  void begin() {
    const BlockedJob* pr = job && job->kind == kJobPredict ? job : nullptr;
    nt = npad / kTile;
    {
      ScopedTimer t(h, CCGP_T_COV, s);
      launch_cov_tiles(s, X, n, d, dv, b0, nb, w.A, w.a_stride, npad, mean_mode, sigma2, tau2, w.ld, w.xpad, w.upad);
      RhsArgs ra{w.A, w.a_stride, npad, n, y, w.ld, job && job->kind >= kJobInverse ? 1 : 0};
      hipLaunchKernelGGL(rhs_rows_kernel, dim3(npad / 16, nb), dim3(256), 0, s, ra);
      if (pr)   // rows npad+128+t = r(x_t)' (Mixed.corr.vec, HX:425-431), one row per test site
        launch_cov_cross_batched(s, pr->Xtest, pr->m, X, n, d, dv, b0, nb, w.A + npad + kTile, w.a_stride,
                                 w.ld);
    }
    g.A = w.A; g.a_stride = w.a_stride; g.npad = npad; g.invd = w.invd;
    g.invd_stride = (size_t)nt * kTile * kTile; g.nt = nt; g.nb = nb; g.ld = w.ld; g.ne = w.ne;
    g.extra_lower = job && job->kind >= kJobInverse ? 1 : 0;
    g.fuse_diag = h->opt_fuse_diag; g.tail_strips = h->opt_tail_strips; g.wide = h->opt_wide_offsets; g.logdet_part = w.z; g.status = status + b0; g.n = n;
    dg.A = w.A; dg.a_stride = w.a_stride; dg.npad = npad; dg.invd = w.invd;
    dg.invd_stride = g.invd_stride; dg.logdet_part = w.z; dg.status = status + b0; dg.nt = nt;
    dg.nb = nb; dg.n = n; dg.ld = w.ld;
    dg.ptol = g.ptol = pivot_tolerance(mean_mode, n);
  }

  // T_ij = A_ij - sum_{k<j} L_ik L_jk' for every tile row of block column j (nothing to do at j = 0)
  void update(int j) {
    if (j == 0) return;
    ScopedTimer t(h, CCGP_T_UPDATE, s);
    g.j = j;
    g.mode = 0;
    launch_gemm(s, g, 0, 1);
  }

  // L_jj, W_j = L_jj^-1, then L_ij = T_ij W_j' for the rows below
  void panel(int j) {
    if (j == 0 || !g.fuse_diag) {   // block column 0 has no update launch; from column 1 on the update's diagonal workgroup does it
      ScopedTimer t(h, CCGP_T_DIAG, s);
      dg.j = j;
      hipLaunchKernelGGL(diag_kernel, dim3(nb), dim3(256), 0, s, dg);
    }
    {
      ScopedTimer t(h, CCGP_T_TRSM, s);
      g.j = j;
      g.mode = 1;
      launch_gemm(s, g, 1, 1);
    }
  }

  void finish() {
    const BlockedJob* pr = job && job->kind == kJobPredict ? job : nullptr;
    {
      ScopedTimer t(h, CCGP_T_SOLVE, s);
      FinishArgs fa{};
      fa.A = w.A; fa.a_stride = w.a_stride; fa.npad = npad; fa.logdet_part = w.z; fa.params = dv.params;
      fa.ldp = dv.ldp; fa.K = dv.K; fa.b0 = b0; fa.nt = nt; fa.n = n; fa.sigma2 = sigma2;
      fa.mode = mean_mode; fa.loglik = loglik; fa.beta = beta; fa.status = status; fa.ld = w.ld;
      fa.s11_out = w.fin; fa.beta_out = w.fin + nb;
      fa.logdet_out = job && job->kind == kJobLogdet ? job->logdet : nullptr;
      hipLaunchKernelGGL(finish_kernel, dim3(nb), dim3(256), 0, s, fa);
      if (pr) {
        PredFinishArgs pa{w.A, w.a_stride, npad, w.ld, n, pr->m, nullptr, 0, 0, w.fin, w.fin + nb, status, b0, pr->S,
                          sigma2, pr->mean, pr->var};
        hipLaunchKernelGGL(predict_finish_kernel, dim3((pr->m + 63) / 64, nb), dim3(64 * kPredFinishWaves), 0, s, pa);
      }
    }
    if (job && job->kind >= kJobInverse) {
      ScopedTimer t(h, CCGP_T_SOLVE, s);
      const int ntiles = nt * (nt + 1) / 2;
      RinvArgs ra{};
      ra.A = w.A; ra.a_stride = w.a_stride; ra.npad = npad; ra.ld = w.ld; ra.nt = nt; ra.n = n; ra.nb = nb;
      ra.wide = h->opt_wide_offsets;
      const dim3 grid(round_up(nb, 8) * ntiles), block(256);
      if (job->kind == kJobInverse) {
        ra.Rinv = job->Rinv;
        hipLaunchKernelGGL(rinv_tile_kernel<false>, grid, block, gemm_lds_bytes<1>(), s, ra);
      } else {
        const int P = dv.K + dv.K * d;
        AlphaArgs aa{w.A, w.a_stride, npad, w.ld, n, w.fin + nb, job->alpha};
        hipLaunchKernelGGL(alpha_kernel, dim3(npad / 64, nb), dim3(64 * kAlphaWaves), 0, s, aa);
        ra.X = X; ra.d = d; ra.K = dv.K; ra.params = dv.params; ra.ldp = dv.ldp; ra.draw0 = b0;
        ra.sigma2 = sigma2; ra.alpha = job->alpha; ra.gpart = job->gpart;
        hipLaunchKernelGGL(rinv_tile_kernel<true>, grid, block, gemm_lds_bytes<1>(), s, ra);
        const int nt64 = npad / 64, ntg = nt64 * (nt64 + 1) / 2;
        GradContractArgs ca{w.A, w.a_stride, w.ld, n, d, dv.K, X, dv.params, dv.ldp, b0, job->gpart, ntg};
        const dim3 cgrid(ntg, 1, nb);
        const size_t clds = grad_contract_lds(d, dv.K);
        if (d <= 4) hipLaunchKernelGGL(grad_contract_kernel<4>, cgrid, block, clds, s, ca);
        else if (d <= 6) hipLaunchKernelGGL(grad_contract_kernel<6>, cgrid, block, clds, s, ca);
        else if (d <= 8) hipLaunchKernelGGL(grad_contract_kernel<8>, cgrid, block, clds, s, ca);
        else hipLaunchKernelGGL(grad_contract_kernel<0>, cgrid, block, clds, s, ca);
        GradReduceArgs ga{job->gpart, ntg, P, dv.K, d, nb, b0, job->Btot, dv.params, dv.ldp, sigma2,
                          status, job->grad};
        hipLaunchKernelGGL(blocked_grad_reduce_kernel, dim3(nb * P), dim3(256), 0, s, ga);
      }
    }
  }

  // The whole factorisation of the chunk as one persistent launch (chol_sched_kernel).  Needs what the launches' default
  // configuration has: whole update tiles and the diagonal workgroup that factorises its block.
  // CCGP_OPT_SCHED 3 (default) chooses by measurement (profiles/r05_experiments.md, n = 4096, same box, chunk sizes 8 ... 512):
  // the scheduler with one workgroup per CU is ahead where the launches lose CUs to their partial last steps and to the late
  // block columns -- 32 ... 128 matrices, the share of one GPU when the 512-point grid is spread over 4 - 8 of them
  // (64 matrices: 26.25 against 26.93 ms) -- and behind by 1.5 - 2.5 % on both sides of that: few matrices (critical path:
  // queue hops instead of launches) and many (whole steps: nothing to recover, and the queue traffic costs 2.4 us per task).
  int sched_mode() const {
    if (h->opt_sched != 3) return h->opt_sched;
    return (nt >= 16 && nb >= 32 && nb <= 128) ? 2 : 0;
  }
  bool scheduled() const {
    // a workgroup serves the queue of the XCD it runs on and matrix b lives on queue b % 8: the whole chip in one partition
    // (8 XCDs x 32 CUs) -- on a partitioned device some queues would have no workgroup and the sweep would end in the abort path
    return sched_mode() != 0 && h->n_cus == 256 && h->opt_fuse_diag && sched::rows(sched::Shape{nt, w.ne, 0}) < 0x7fff &&
           (size_t)nb * (size_t)sched::tasks_per_matrix(sched::Shape{nt, w.ne, 0}) < 0x7fffffffull;
  }
  void sweep_scheduled() {
    ScopedTimer t(h, CCGP_T_SWEEP, s);
    const sched::Shape shp{nt, w.ne, g.extra_lower};
    const long tpm = sched::tasks_per_matrix(shp), total = tpm * nb;
    unsigned long long* slots = reinterpret_cast<unsigned long long*>(w.sched);
    int* counters = reinterpret_cast<int*>(slots + (size_t)nb * sched::tasks_per_matrix(sched::Shape{nt, w.ne, 0}));
    int* ctrl = counters + (size_t)nb * sched::counters_per_matrix(shp);
    int* cu_seen = ctrl + kSchedCtrlInts;
    h->sched_prof_dev = reinterpret_cast<unsigned long long*>(cu_seen + 8 * 128 + ((8 * 128 + kSchedCtrlInts + (size_t)nb * sched::counters_per_matrix(shp)) & 1));
    SchedInitArgs ia{ctrl, slots, counters, cu_seen, nb, nt, w.ne, g.extra_lower, tpm, total};
    const int init_blocks = (int)std::min<long>(2048, (total + 255) / 256 + 8);
    hipLaunchKernelGGL(sched_init_kernel, dim3(init_blocks), dim3(256), 0, s, ia);
    SchedArgs a{};
    a.g = g; a.g.j = 0; a.g.mode = 0; a.g.n_s1 = 0; a.g.tail_s = 2;
    a.ctrl = ctrl; a.slots = slots; a.counters = counters; a.cu_seen = cu_seen;
    a.policy = h->opt_sched_policy;
    a.backlog_min = h->sched_backlog_min > 0 ? h->sched_backlog_min : std::max(1, h->n_cus / 8);
    a.prof = h->sched_prof_dev;
    a.timeout_10ns = h->sched_timeout_ms >= 40000 ? 4000000000u : (unsigned)h->sched_timeout_ms * 100000u;
    const int wgs = h->n_cus * (sched_mode() == 1 ? 2 : 1);
    h->sched_prof_wgs = wgs < kSchedProfWgs ? wgs : kSchedProfWgs;
    hipLaunchKernelGGL(chol_sched_kernel, dim3(wgs), dim3(256), gemm_lds_bytes<1>(), s, a);
    hipLaunchKernelGGL(sched_check_kernel, dim3(1), dim3(256), 0, s, ctrl, status + b0, nb);
  }

  void run_all() {
    begin();
    if (scheduled()) {
      sweep_scheduled();
    } else {
      for (int j = 0; j < nt; ++j) {
        update(j);
        panel(j);
      }
    }
    finish();
  }
};

// LDS of the gradient contraction (grad_contract_kernel) must fit a workgroup's share
bool blocked_grad_supported(int d, int K) { return grad_contract_lds(d, K) <= (size_t)kLdsBytes - 64; }

// partial sums of the gradient contraction per matrix: one per lower 64 x 64 tile
size_t blocked_grad_partials(int npad) { const size_t nt64 = (size_t)npad / 64; return nt64 * (nt64 + 1) / 2; }

void blocked_loglik(ccgp_handle* h, const double* X, int n, int d, const double* y, DrawView dv,
                    int b0, int nb, int npad, double sigma2, int mean_mode, double tau2,
                    BlockedWs w, double* loglik, double* beta, int* status, const BlockedJob* job) {
  static unsigned long long attr_mask = 0;
  once_per_device(attr_mask, [] {
    raise_lds_limit((const void*)chol_update_kernel, "chol_update_kernel");
    raise_lds_limit((const void*)chol_update_s2_kernel, "chol_update_s2_kernel");
    raise_lds_limit((const void*)chol_trsm_kernel, "chol_trsm_kernel");
    raise_lds_limit((const void*)chol_sched_kernel, "chol_sched_kernel");
    raise_lds_limit((const void*)rinv_tile_kernel<false>, "rinv_tile_kernel<false>");
    raise_lds_limit((const void*)rinv_tile_kernel<true>, "rinv_tile_kernel<true>");
    raise_lds_limit((const void*)grad_contract_kernel<4>, "grad_contract_kernel<4>");
    raise_lds_limit((const void*)grad_contract_kernel<6>, "grad_contract_kernel<6>");
    raise_lds_limit((const void*)grad_contract_kernel<8>, "grad_contract_kernel<8>");
    raise_lds_limit((const void*)grad_contract_kernel<0>, "grad_contract_kernel<0>");
  });
  GroupRun r{};
  r.h = h; r.s = h->stream; r.X = X; r.n = n; r.d = d; r.y = y; r.dv = dv; r.b0 = b0; r.nb = nb;
  r.npad = npad; r.sigma2 = sigma2; r.mean_mode = mean_mode; r.tau2 = tau2; r.w = w;
  r.loglik = loglik; r.beta = beta; r.status = status; r.job = job;
  // All matrices of the chunk advance together on one stream.  Splitting the batch into groups on
  // separate streams was measured and rejected twice (profiles/r01c_strip_selection.md): free-running
  // groups (+4 % at 2 groups, slower at 4+) and an event-ordered two-group ping-pong that hides
  // diag / trsm underneath the other half's update (33.9 vs 33.8 ms: the half-size update launches
  // lose to wave quantisation what the overlap gains).
  r.run_all();
}

// ---- prediction from a kept factor set ----------------------------------------------
// The factor of every draw stays in HBM exactly as the sweep left it (lower tiles L, inverted diagonal
// blocks W_j, the rows z_y = L^-1 y and z_1 = L^-1 1, beta, 1'R^-1 1): what Metro caches per accepted draw
// as R.Inv (HX:515-525) and predict.post reads back (HX:655-665).  A new set of test sites then costs
// only the forward substitution of its m cross-correlation rows: the SAME update / trsm tile kernels,
// launched over the extra rows alone (rows_only), so the result is bit-identical to a full
// ccgp_predict_batch sweep -- same kernels, same k order -- at O(m n^2) instead of O(n^3).
// Served in chunks of draws [s0, s0 + ns): the scratch rows E are sized under the handle's workspace limit.
void blocked_predict_from_factors(ccgp_handle* h, const BlockedWs& w, int n, int npad, int S, int s0, int ns, double* E,
                                  size_t e_stride, int lde, int m, const int* status, double sigma2,
                                  double* mean, double* var) {
  static unsigned long long attr_mask = 0;
  once_per_device(attr_mask, [] {
    raise_lds_limit((const void*)chol_update_kernel, "chol_update_kernel");
    raise_lds_limit((const void*)chol_update_s2_kernel, "chol_update_s2_kernel");
    raise_lds_limit((const void*)chol_trsm_kernel, "chol_trsm_kernel");
  });
  const int nt = npad / kTile, ne = lde / kTile;
  hipStream_t s = h->stream;
  GemmArgs g{};
  g.invd_stride = (size_t)nt * kTile * kTile;
  // draws [s0, s0 + ns) of the set's S: the factor arrays are offset, E holds this chunk's rows only
  g.A = w.A + (size_t)s0 * w.a_stride; g.a_stride = w.a_stride; g.npad = npad; g.invd = w.invd + (size_t)s0 * g.invd_stride;
  g.nt = nt; g.nb = ns; g.ld = w.ld; g.ne = ne;
  g.extra_lower = 0; g.E = E; g.e_stride = e_stride; g.lde = lde; g.rows_only = 1;
  const int nb8 = round_up(ns, 8);
  for (int j = 0; j < nt; ++j) {
    g.j = j;
    if (j > 0) {
      ScopedTimer t(h, CCGP_T_UPDATE, s);
      g.mode = 0;
      // few rows: two half-width strips double the workgroups of a launch that cannot fill the chip
      launch_gemm(s, g, 0, nb8 * ne < 256 ? 2 : 1);
    }
    {
      ScopedTimer t(h, CCGP_T_TRSM, s);
      g.mode = 1;
      launch_gemm(s, g, 1, 1);
    }
  }
  ScopedTimer t(h, CCGP_T_SOLVE, s);
  PredFinishArgs pa{g.A, w.a_stride, npad, w.ld, n, m, E, e_stride, lde, w.fin + s0, w.fin + S + s0, status, s0, S,
                    sigma2, mean, var};
  hipLaunchKernelGGL(predict_finish_kernel, dim3((m + 63) / 64, ns), dim3(64 * kPredFinishWaves), 0, s, pa);
}

}  // namespace ccgp
