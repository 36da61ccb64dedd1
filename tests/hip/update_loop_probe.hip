// Where do the 15 % of non-MFMA cycles of the trailing-update inner loop go?  The loop of
// gemm_accumulate<1, false> (csrc/blocked.hip) rebuilt with its ingredients switchable:
//   LDS   fragment ds_reads of every k-step (otherwise register operands)
//   BAR   the stage barrier
//   DMA   the eight global_load_lds of the next stage (0: none, 1: from a 64 KB window that stays in L2,
//         2: streaming through a panel as the real kernel does)
// for one and two workgroups per CU.  Prints TFLOP/s and the implied cycles per MFMA at 2.4 GHz.
// Build: hipcc --offload-arch=gfx950 -O3 update_loop_probe.hip -o update_loop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int kTile = 128, BK = 16, STAGE = BK * kTile * 2;

template <bool LDS, bool BAR, int DMA>
__global__ __launch_bounds__(256, 2) void probe(const double* __restrict__ panels, size_t panel_stride, int ld,
                                                int nk, double* out, int share) {
  extern __shared__ double smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = (wave >> 1) * 64, col0 = (wave & 1) * 64;
  const int l15 = lane & 15, l4 = lane >> 4, sw = l4 & 1;
  d4 acc[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = d4{0, 0, 0, 0};
  for (int i = tid; i < 2 * STAGE; i += 256) smem[i] = 1e-3 * (i & 255);
  __syncthreads();

  const double* P = panels + (size_t)(blockIdx.x / share) * panel_stride;
  const double* Q = P + kTile;
  const int psrc = ((((lane >> 3) ^ (wave & 1)) << 4) + ((lane & 7) << 1));
  const double* pP = P + psrc + (size_t)wave * ld;
  const double* pQ = Q + psrc + (size_t)wave * ld;
  const size_t step = (size_t)4 * ld;
  const double* pP0 = pP;
  const double* pQ0 = pQ;
  auto issue = [&](int stage, int kt) {
    if constexpr (DMA != 0) {
      double* Ps_ = smem + stage * STAGE + wave * kTile;
      double* Qs_ = Ps_ + BK * kTile;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pP + q * step),
                                         (__attribute__((address_space(3))) void*)(Ps_ + 4 * q * kTile), 16, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pQ + q * step),
                                         (__attribute__((address_space(3))) void*)(Qs_ + 4 * q * kTile), 16, 0, 0);
      if (DMA == 2) { pP += 4 * step; pQ += 4 * step; }
      else if ((kt & 1) == 0) { pP += 4 * step; pQ += 4 * step; } else { pP = pP0; pQ = pQ0; }
    }
  };
  double pfA[4], qfA[4], pfB[4], qfB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { pfA[i] = 1.0 + lane * 1e-3 + i; qfA[i] = 0.5 - lane * 1e-3 + i; pfB[i] = pfA[i] + 1; qfB[i] = qfA[i] - 1; }
#define LOADF(PF, QF, STG, KK)                                                                   \
  do {                                                                                          \
    if constexpr (LDS) {                                                                        \
      const double* Ps_ = smem + (STG) * STAGE;                                                 \
      const double* Qs_ = Ps_ + BK * kTile;                                                     \
      _Pragma("unroll") for (int y = 0; y < 4; ++y)                                             \
          PF[y] = Ps_[((KK) * 4 + l4) * kTile + ((((row0 >> 4) + y) ^ sw) << 4) + l15];         \
      _Pragma("unroll") for (int x = 0; x < 4; ++x)                                             \
          QF[x] = Qs_[((KK) * 4 + l4) * kTile + ((((col0 >> 4) + x) ^ sw) << 4) + l15];         \
    }                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  } while (0)
#define MFMAS(PF, QF, X0, X1)                                                                    \
  do {                                                                                          \
    _Pragma("unroll") for (int x = (X0); x < (X1); ++x)                                         \
      _Pragma("unroll") for (int y = 0; y < 4; ++y)                                             \
          acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(QF[x], PF[y], acc[x][y], 0, 0, 0);   \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  } while (0)
  issue(0, 0);
  __syncthreads();
  if (nk > 1) issue(1, 1);
  LOADF(pfA, qfA, 0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int stg = kt & 1;
    MFMAS(pfA, qfA, 0, 1);
    LOADF(pfB, qfB, stg, 1);
    MFMAS(pfA, qfA, 1, 4);
    MFMAS(pfB, qfB, 0, 1);
    LOADF(pfA, qfA, stg, 2);
    MFMAS(pfB, qfB, 1, 4);
    MFMAS(pfA, qfA, 0, 1);
    LOADF(pfB, qfB, stg, 3);
    MFMAS(pfA, qfA, 1, 4);
    if (kt + 1 < nk) {
      if constexpr (BAR) __syncthreads();
      if (kt + 2 < nk) issue(stg, kt + 2);
    }
    MFMAS(pfB, qfB, 0, 1);
    if (kt + 1 < nk) LOADF(pfA, qfA, stg ^ 1, 0);
    MFMAS(pfB, qfB, 1, 4);
  }
  double s = 0;
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) s += acc[x][y][0] + acc[x][y][1] + acc[x][y][2] + acc[x][y][3];
  out[(size_t)blockIdx.x * 256 + tid] = s;
}


// The same loop with the fragment reads and the DMA requests dealt out one per MFMA (each MFMA occupies its pipe for
// 64 cycles, during which the wave can issue something else for free) and all LDS / global address arithmetic on
// the scalar unit (wave index through readfirstlane; global address = SGPR base + 32-bit lane offset).
template <bool BAR, int DMA, bool F128>
__global__ __launch_bounds__(256, 2) void probe2(const double* __restrict__ panels, size_t panel_stride, int ld,
                                                 int nk, double* out, int share) {
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

  const char* gP = (const char*)(panels + (size_t)(blockIdx.x / share) * panel_stride) + (size_t)wave * ld * 8;
  const char* gQ = gP + kTile * 8;
  const unsigned lane_off = (unsigned)(((((lane >> 3) ^ (wave & 1)) << 4) + ((lane & 7) << 1)) * 8);
  const size_t stepB = (size_t)4 * ld * 8;
  // DMA request r = 0..7 of a stage: r < 4 -> P rows, else Q rows
  // DMA == 3: buffer_load ... lds: lane offset in ONE 32-bit VGPR, everything else in SGPRs
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(panels + (size_t)(blockIdx.x / share) * panel_stride), 0, 0x7fffffff, 0x00020000);
  unsigned bofs = (unsigned)wave * ld * 8;   // scalar byte offset of this wave's first P row
  const unsigned lane_lin = F128 ? (unsigned)lane * 16 : lane_off;
  auto issue_one = [&](int stage, int r) {
    if constexpr (DMA == 3) {
      double* dst = smem + stage * STAGE + (r >> 2) * (BK * kTile) + wave * kTile + 4 * (r & 3) * kTile;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, lane_lin,
                                               bofs + (unsigned)((r & 3) * stepB) + (r >> 2) * kTile * 8, 0, 0);
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

__global__ void fill(double* p, size_t n) {
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    p[i] = 1.0 + 1e-6 * (double)((i * 2654435761u) & 0xfffff);
}

template <bool LDS, bool BAR, int DMA>
static void run(const char* name, const double* panels, size_t panel_stride, int ld, int nk, double* out,
                int share = 1) {
  hipFuncSetAttribute((const void*)probe<LDS, BAR, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int per_cu = 1; per_cu <= 2; ++per_cu) {
    const int grid = 256 * per_cu;
    float ms = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {   // best of four; the first also warms the clocks
      hipEventRecord(e0);
      hipLaunchKernelGGL((probe<LDS, BAR, DMA>), dim3(grid), dim3(256), 2 * STAGE * 8, 0, panels, panel_stride, ld, nk, out, share);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float t; hipEventElapsedTime(&t, e0, e1);
      if (t < ms) ms = t;
    }
    const double mfmas = (double)nk * 64;
    printf("%-34s WG/CU=%d  %7.3f ms  %5.1f TFLOP/s  cycles/MFMA@2.4GHz = %.1f\n", name, per_cu, ms,
           (double)grid * 4 * mfmas * 2048.0 / ms / 1e9, 2.4e9 * ms * 1e-3 / (mfmas * per_cu));
  }
}

template <bool BAR, int DMA, bool F128>
static void run2(const char* name, const double* panels, size_t panel_stride, int ld, int nk, double* out,
                 int share = 1) {
  hipFuncSetAttribute((const void*)probe2<BAR, DMA, F128>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int per_cu = 1; per_cu <= 2; ++per_cu) {
    const int grid = 256 * per_cu;
    float ms = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL((probe2<BAR, DMA, F128>), dim3(grid), dim3(256), 2 * STAGE * 8, 0, panels, panel_stride, ld, nk, out, share);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float t; hipEventElapsedTime(&t, e0, e1);
      if (t < ms) ms = t;
    }
    const double mfmas = (double)nk * 64;
    printf("%-34s WG/CU=%d  %7.3f ms  %5.1f TFLOP/s  cycles/MFMA@2.4GHz = %.1f\n", name, per_cu, ms,
           (double)grid * 4 * mfmas * 2048.0 / ms / 1e9, 2.4e9 * ms * 1e-3 / (mfmas * per_cu));
  }
}

int main() {
  // panels: 512 workgroups x (ld = 256 doubles wide: P in columns 0..127, Q in 128..255) x K rows
  const int nk = 2048, ld = 256;   // K = 32768: 3.5 ms of MFMAs per workgroup alone on its CU
  const size_t panel_stride = (size_t)ld * (nk * BK + 64);
  double *panels, *out;
  if (hipMalloc(&panels, 512 * panel_stride * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, panels, 512 * panel_stride);
  hipMalloc(&out, 512 * 256 * 8);
  for (int i = 0; i < 100; ++i)   // ~0.4 s of MFMAs before anything is timed
    hipLaunchKernelGGL((probe<false, false, 0>), dim3(512), dim3(256), 2 * STAGE * 8, 0, panels, panel_stride, ld, nk, out, 1);
  hipDeviceSynchronize();
  run<false, false, 0>("mfma only", panels, panel_stride, ld, nk, out);
  run<true, false, 0>("+ lds fragments", panels, panel_stride, ld, nk, out);
  run<true, true, 0>("+ lds + barrier", panels, panel_stride, ld, nk, out);
  run<false, true, 0>("barrier only", panels, panel_stride, ld, nk, out);
  run<true, true, 1>("+ lds + barrier + dma (L2 window)", panels, panel_stride, ld, nk, out);
  run<true, true, 2>("+ lds + barrier + dma (streaming)", panels, panel_stride, ld, nk, out);
  run<true, true, 2>("+ ... dma (streaming, 16 share)", panels, panel_stride, ld, nk, out, 16);
  run<false, true, 2>("barrier + dma (16 share), no lds", panels, panel_stride, ld, nk, out, 16);
  run2<true, 0, false>("interleaved: lds + barrier", panels, panel_stride, ld, nk, out);
  run2<true, 2, false>("interleaved: + dma (16 share)", panels, panel_stride, ld, nk, out, 16);
  run2<true, 0, true>("interleaved: lds b128 + barrier", panels, panel_stride, ld, nk, out);
  run2<true, 2, true>("il: lds b128 + dma (16 share)", panels, panel_stride, ld, nk, out, 16);
  run2<true, 3, false>("il: lds + buffer dma (16 share)", panels, panel_stride, ld, nk, out, 16);
  run2<true, 3, true>("il: lds b128 + buffer dma (16 sh)", panels, panel_stride, ld, nk, out, 16);
  // sustained: do the clocks hold over the length of a whole 512-matrix sweep (~200 ms)?
  {
    hipFuncSetAttribute((const void*)probe2<true, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE * 8);
    hipEvent_t e[9];
    for (int i = 0; i < 9; ++i) hipEventCreate(&e[i]);
    for (int which = 0; which < 2; ++which) {
      hipDeviceSynchronize();
      hipEventRecord(e[0]);
      for (int seg = 0; seg < 8; ++seg) {
        for (int i = 0; i < 8; ++i) {
          if (which == 0) hipLaunchKernelGGL((probe<false, false, 0>), dim3(512), dim3(256), 2 * STAGE * 8, 0, panels, panel_stride, ld, nk, out, 1);
          else hipLaunchKernelGGL((probe2<true, 3, true>), dim3(512), dim3(256), 2 * STAGE * 8, 0, panels, panel_stride, ld, nk, out, 16);
        }
        hipEventRecord(e[seg + 1]);
      }
      hipEventSynchronize(e[8]);
      printf("sustained %s, TFLOP/s per ~57 ms segment:", which == 0 ? "mfma only" : "full loop (il b128 + buffer dma)");
      for (int seg = 0; seg < 8; ++seg) {
        float t; hipEventElapsedTime(&t, e[seg], e[seg + 1]);
        printf(" %.1f", 8.0 * 512 * 4 * (double)nk * 64 * 2048.0 / t / 1e9);
      }
      printf("\n");
    }
  }
  return 0;
}
