// Does the LAYOUT of the panels in HBM matter to the update loop?  The loop of tile_accumulate_il (csrc/blocked.hip)
// over 512 matrices x 16 tiles, K = 2048 (the shape of block column 16 of the n = 4096 sweep), every tile streaming
// its OWN row panel (128 rows x K) and sharing the column operand with the 15 other tiles of its matrix:
//   column-major, ld = 4352: a stage is 16 pieces of 1 KiB, 34 KiB apart (what the workspace holds today)
//   tile-row-major, ld = 128: a stage is 16 KiB contiguous, a panel one 2 MiB stream
// Build: hipcc --offload-arch=gfx950 -O3 panel_layout_probe.hip -o panel_layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int kTile = 128, BK = 16, STAGE = BK * kTile * 2;

// The same loop with the fragment reads and the DMA requests dealt out one per MFMA (each MFMA occupies its pipe for
// 64 cycles, during which the wave can issue something else for free) and all LDS / global address arithmetic on
// the scalar unit (wave index through readfirstlane; global address = SGPR base + 32-bit lane offset).
template <bool BAR, int DMA, bool F128>
__global__ __launch_bounds__(256, 2) void probe3(const double* __restrict__ A, size_t a_stride, size_t row_stride,
                                                 int ld, int nk, double* out, int tiles) {
  extern __shared__ double smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = (wave >> 1) * 64, col0 = (wave & 1) * 64;
  const int l15 = lane & 15, l4 = lane >> 4, sw = l4 & 1;
  d4 acc[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = d4{0, 0, 0, 0};
  for (int i = tid; i < 2 * STAGE; i += 256) smem[i] = 1e-3 * (i & 255);
  __syncthreads();

  const int L = blockIdx.x, b = (L / (8 * tiles)) * 8 + (L & 7), ti = 1 + (L >> 3) % tiles;
  const double* Pb = A + (size_t)b * a_stride + (size_t)ti * row_stride;   // row panel of tile row ti
  const double* Qb = A + (size_t)b * a_stride;                             // column operand: tile row 0 (shared)
  const char* gP = (const char*)Pb + (size_t)wave * ld * 8;
  const char* gQ = (const char*)Qb + (size_t)wave * ld * 8;
  const unsigned lane_off = (unsigned)(((((lane >> 3) ^ (wave & 1)) << 4) + ((lane & 7) << 1)) * 8);
  const size_t stepB = (size_t)4 * ld * 8;
  // DMA request r = 0..7 of a stage: r < 4 -> P rows, else Q rows
  // DMA == 3: buffer_load ... lds: lane offset in ONE 32-bit VGPR, everything else in SGPRs
  __amdgpu_buffer_rsrc_t rsrcP = __builtin_amdgcn_make_buffer_rsrc((void*)Pb, 0, -1, 0x00020000);
  __amdgpu_buffer_rsrc_t rsrcQ = __builtin_amdgcn_make_buffer_rsrc((void*)Qb, 0, -1, 0x00020000);
  unsigned bofs = (unsigned)wave * ld * 8;   // scalar byte offset of this wave's first P row
  const unsigned lane_lin = F128 ? (unsigned)lane * 16 : lane_off;
  auto issue_one = [&](int stage, int r) {
    if constexpr (DMA == 3) {
      double* dst = smem + stage * STAGE + (r >> 2) * (BK * kTile) + wave * kTile + 4 * (r & 3) * kTile;
      if (r < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcP, (__attribute__((address_space(3))) void*)dst, 16, lane_lin,
                                                          bofs + (unsigned)((r & 3) * stepB), 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcQ, (__attribute__((address_space(3))) void*)dst, 16, lane_lin,
                                                    bofs + (unsigned)((r & 3) * stepB), 0, 0);
    } else if constexpr (DMA != 0) {
      double* dst = smem + stage * STAGE + (r >> 2) * (BK * kTile) + wave * kTile + 4 * (r & 3) * kTile;
      const char* src = ((r >> 2) ? gQ : gP) + (size_t)(r & 3) * stepB + lane_off;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };
  auto advance = [&]() { gP += 4 * stepB; gQ += 4 * stepB; bofs += 4 * (unsigned)stepB; };

  // fragment element offsets (doubles) inside a stage for k-step 0
  int pofs[4], qofs[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    pofs[t] = l4 * kTile + ((((row0 >> 4) + t) ^ sw) << 4) + l15;
    qofs[t] = BK * kTile + l4 * kTile + ((((col0 >> 4) + t) ^ sw) << 4) + l15;
  }
  // F128: sub-tile t of a wave holds rows row0 + 4 * l15 + t, so a lane's four fragments are 32 contiguous bytes
  const int pofs128 = l4 * kTile + row0 + 4 * l15, qofs128 = BK * kTile + l4 * kTile + col0 + 4 * l15;
  typedef double d2 __attribute__((ext_vector_type(2)));
  double pfA[4], qfA[4], pfB[4], qfB[4];
#define SB __builtin_amdgcn_sched_barrier(0)
#define LOAD1(PF, QF, STG, KK, I)                                                                 \
  do {                                                                                           \
    const double* St_ = smem + (STG) * STAGE + (KK) * 4 * kTile;                                 \
    if constexpr (F128) {                                                                        \
      if ((I) == 0) { d2 v_ = *(const d2*)(St_ + pofs128); PF[0] = v_[0]; PF[1] = v_[1]; }       \
      if ((I) == 1) { d2 v_ = *(const d2*)(St_ + pofs128 + 2); PF[2] = v_[0]; PF[3] = v_[1]; }   \
      if ((I) == 2) { d2 v_ = *(const d2*)(St_ + qofs128); QF[0] = v_[0]; QF[1] = v_[1]; }       \
      if ((I) == 3) { d2 v_ = *(const d2*)(St_ + qofs128 + 2); QF[2] = v_[0]; QF[3] = v_[1]; }   \
    } else {                                                                                     \
      if ((I) < 4) PF[(I) & 3] = St_[pofs[(I) & 3]]; else QF[(I) & 3] = St_[qofs[(I) & 3]];      \
    }                                                                                            \
  } while (0)
// one k-step: 16 MFMAs; after MFMA i < 8 the i-th fragment of the NEXT k-step is requested (LD), and in the stage's
// last k-step also the i-th DMA request of the stage after next (DM)
#define KSTEP(PFc, QFc, PFn, QFn, STGn, KKn, LD, DM, STGd)                                        \
  do {                                                                                           \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                             \
      acc[i >> 2][i & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(QFc[i >> 2], PFc[i & 3], acc[i >> 2][i & 3], 0, 0, 0); \
      if ((LD) && i < 8) LOAD1(PFn, QFn, STGn, KKn, i);                                          \
      if ((DM) && i < 8) issue_one(STGd, i);                                                     \
      SB;                                                                                        \
    }                                                                                            \
  } while (0)

  for (int r = 0; r < 8; ++r) issue_one(0, r);
  advance();
  __syncthreads();
  if (nk > 1) { for (int r = 0; r < 8; ++r) issue_one(1, r); advance(); }
  for (int i = 0; i < 8; ++i) LOAD1(pfA, qfA, 0, 0, i);
  SB;
#define STAGE_BODY(MORE, REFILL)                                                                  \
  do {                                                                                           \
    const int stg = kt & 1;                                                                      \
    KSTEP(pfA, qfA, pfB, qfB, stg, 1, true, false, 0);                                           \
    KSTEP(pfB, qfB, pfA, qfA, stg, 2, true, false, 0);                                           \
    KSTEP(pfA, qfA, pfB, qfB, stg, 3, true, false, 0);                                           \
    if (MORE) {                                                                                  \
      if constexpr (BAR) __syncthreads();                                                        \
      SB;                                                                                        \
    }                                                                                            \
    KSTEP(pfB, qfB, pfA, qfA, stg ^ 1, 0, MORE, REFILL, stg);                                    \
    if (REFILL) advance();                                                                       \
  } while (0)
  int kt = 0;
  for (; kt < nk - 2; ++kt) STAGE_BODY(true, true);
  if (nk >= 2) { STAGE_BODY(true, false); ++kt; }
  STAGE_BODY(false, false);
  double s = 0;
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) s += acc[x][y][0] + acc[x][y][1] + acc[x][y][2] + acc[x][y][3];
  out[(size_t)blockIdx.x * 256 + tid] = s;
}


// mode 0: 1 + tiny (few mantissa bits toggle); mode 1: full-entropy mantissas in (-1, 1); mode 2: NaN
__global__ void fill(double* p, size_t n, int mode) {
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    unsigned long long h = i * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    const double u = (double)(h >> 11) * (1.0 / 9007199254740992.0);   // [0, 1), 53 random bits
    p[i] = mode == 0 ? 1.0 + 1e-6 * (double)((i * 2654435761u) & 0xfffff)
                     : (mode == 1 ? 2.0 * u - 1.0 : __longlong_as_double(0x7ff8000000000000LL));
  }
}

int main() {
  const int nmat = 512, tiles = 16, K = 2048, nk = K / BK, nrowt = tiles + 1;
  const size_t per_matrix = (size_t)nrowt * kTile * K;   // doubles, either layout
  double *A, *out;
  if (hipMalloc(&A, nmat * per_matrix * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&out, (size_t)nmat * tiles * 256 * 8);
  hipFuncSetAttribute((const void*)probe3<true, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = nmat * tiles;
  for (int mode = 0; mode < 3; ++mode)
  for (int layout = 0; layout < 2; ++layout) {
    if (layout == 0) { hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, A, nmat * per_matrix, mode); hipDeviceSynchronize(); }
    const int ld = layout == 0 ? nrowt * kTile : kTile;
    const size_t row_stride = layout == 0 ? (size_t)kTile : (size_t)kTile * K;
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL((probe3<true, 3, true>), dim3(grid), dim3(256), 2 * STAGE * 8, 0, A, per_matrix, row_stride, ld, nk, out, tiles);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float t; hipEventElapsedTime(&t, e0, e1);
      if (t < best) best = t;
    }
    const double flop = (double)grid * 4 * nk * 64 * 2048.0, bytes = (double)nmat * per_matrix * 8;
    printf("%s  %-16s ld = %4d  %7.3f ms  %5.1f TFLOP/s  panel bytes once each: %.2f TB/s\n",
           mode == 0 ? "data 1+tiny" : (mode == 1 ? "data random" : "data NaN   "), layout == 0 ? "column-major" : "tile-row-major", ld, best, flop / best / 1e9, bytes / best / 1e9);
  }
  return 0;
}
