// Do fp64 VALU instructions of one wave overlap with the fp64 MFMAs of ANOTHER wave on the same SIMD?
// The datasheet gives both instruction classes the same 78.6 TFLOP/s, and rounds 1 - 3 observed that every attempt to run
// the VALU-bound covariance generation beside the MFMA-bound trailing update "stretched both".  This probe asks the
// hardware directly, because the answer decides whether generating a covariance tile inside the update workgroup that
// consumes it can hide behind the other resident workgroup's MFMAs or must be paid in full.
//   One workgroup of 256 (1 + NV) threads per CU: waves 0-3 (one per SIMD) run a v_mfma_f64_16x16x4 loop with four
//   independent accumulators, the other 4 NV waves (NV per SIMD) run an fp64 VALU loop (8 independent chains).
//   Three launches: MFMA waves only, VALU waves only, both.  If "both" takes max(a, b) the classes overlap; if it takes
//   a + b they share the pipe.  Roles are WAVE-UNIFORM branches (readfirstlane): with a divergent role every wave
//   walks both loops, one of them under EXEC = 0, and the probe reads "a + b" whatever the hardware does.
// Cycles and clock are measured in the kernel (s_memtime / s_memrealtime) per role; HW_ID tells whether wave w and the
// VALU waves w + 4 q really sit on the same SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
struct Stamp { unsigned long long cyc, rt, from_entry; unsigned hwid, xcc; };

// mode bit 0: MFMA waves work, bit 1: VALU waves work
// VMIX: 0 = v_fma_f64 (one SGPR source) only, 1 = fma + add + an LDS table read (the shape of exp_cov)
template <int NV, int VMIX>
__global__ __launch_bounds__(256 * (1 + NV)) void overlap_kernel(double* out, Stamp* st, int it_mfma, int it_valu, int mode, double c) {
  __shared__ double tab[256];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // stamped at ENTRY, before any vector instruction: a VALU wave that is starved by the MFMA waves shows it as cycles
  const unsigned long long tin = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (tid < 256) tab[tid] = 1.0 + 1e-6 * tid;
  __syncthreads();
  double res = 0.0;
  unsigned long long t0 = 0, t1 = 0, r0 = 0, r1 = 0;
  if (wave < 4) {
    if (mode & 1) {
      d4 acc[4];
      for (int i = 0; i < 4; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
      // full-entropy operands (the clock the chip holds depends on the data)
      unsigned long long z = (tid + 1) * 0x9E3779B97F4A7C15ull;
      z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
      const double a = __longlong_as_double(0x3ff0000000000000ull | (z >> 12)) - 1.5;
      z *= 0x94D049BB133111EBull; z ^= z >> 31;
      const double b = __longlong_as_double(0x3ff0000000000000ull | (z >> 12)) - 1.5;
      t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
      for (int it = 0; it < it_mfma; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      }
      t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
      for (int i = 0; i < 4; ++i) res += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
  } else {
    if (mode & 2) {
      double v[8], bb[8];
      for (int i = 0; i < 8; ++i) { v[i] = 1.0 + 1e-3 * (lane + i); bb[i] = 1e-9 * (i + 1 + lane); }
      t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
      for (int it = 0; it < it_valu; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (VMIX <= 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[i]) : "s"(c), "v"(bb[i]));
          if (VMIX == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(bb[i]) : "v"(v[i]));
          if (VMIX == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(reinterpret_cast<unsigned*>(&v[i])[0]) : "v"(lane));
          if (VMIX == 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(reinterpret_cast<float*>(&v[i])[0]) : "v"(reinterpret_cast<float*>(&bb[i])[0]));
        }
        if (VMIX == 1) v[it & 7] += tab[(lane + it) & 255];
      }
      t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
      for (int i = 0; i < 8; ++i) res += v[i] + bb[i];
    }
  }
  out[blockIdx.x * 256 * (1 + NV) + tid] = res;
  if (lane == 0) {
    Stamp& s = st[blockIdx.x * 4 * (1 + NV) + wave];
    s.cyc = t1 - t0;
    s.rt = r1 - r0;
    s.from_entry = t1 ? t1 - tin : 0;
    s.hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID: simd_id [5:4], cu_id [11:8], sh, se
    s.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // XCC_ID
  }
}

struct Res { double ms, cyc_mfma, cyc_valu, ghz; int shared, wgs; double entry_mfma, entry_valu; };

template <int NV, int VMIX>
static Res run(int mode, int it_mfma, int it_valu, double* d, Stamp* dst) {
  const int grid = 256, W = 4 * (1 + NV);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int r = 0; r < 40; ++r)
    hipLaunchKernelGGL((overlap_kernel<NV, VMIX>), dim3(grid), dim3(64 * W), 0, 0, d, dst, it_mfma, it_valu, mode, 1.0000001);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((overlap_kernel<NV, VMIX>), dim3(grid), dim3(64 * W), 0, 0, d, dst, it_mfma, it_valu, mode, 1.0000001);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<Stamp> st((size_t)grid * W);
  (void)hipMemcpy(st.data(), dst, sizeof(Stamp) * st.size(), hipMemcpyDeviceToHost);
  std::vector<double> cm, cv, gz, em, ev;
  int shared = 0;
  for (int g = 0; g < grid; ++g) {
    bool all = true;
    for (int w = 0; w < W; ++w) {
      const Stamp& q = st[(size_t)g * W + w];
      all = all && ((q.hwid >> 4) & 3) == ((st[(size_t)g * W + (w & 3)].hwid >> 4) & 3);
      if (q.cyc == 0) continue;
      (w < 4 ? cm : cv).push_back((double)q.cyc);
      (w < 4 ? em : ev).push_back((double)q.from_entry);
      if (q.rt) gz.push_back((double)q.cyc / (double)q.rt * 0.1);
    }
    shared += all;
  }
  auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  Res r{ms, med(cm), med(cv), med(gz), shared, grid, med(em), med(ev)};
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return r;
}

template <int NV, int VMIX>
static void scenario(const char* what, int it_mfma, int it_valu, double* d, Stamp* dst) {
  const Res a = run<NV, VMIX>(1, it_mfma, it_valu, d, dst), b = run<NV, VMIX>(2, it_mfma, it_valu, d, dst),
            c = run<NV, VMIX>(3, it_mfma, it_valu, d, dst);
  const double per_it = VMIX == 1 ? 17.0 : 8.0;
  printf("\n%s: 1 MFMA wave + %d VALU wave(s) per SIMD; %d of %d workgroups have wave w and waves w + 4q on one SIMD\n", what, NV, c.shared, c.wgs);
  printf("  %-18s %8s %22s %26s %7s\n", "launch", "ms", "cycles per MFMA (wave)", "SIMD cycles per VALU inst", "GHz");
  printf("  %-18s %8.3f %22.2f %26s %7.3f\n", "MFMA waves alone", a.ms, a.cyc_mfma / it_mfma / 4.0, "-", a.ghz);
  printf("  %-18s %8.3f %22s %26.2f %7.3f\n", "VALU waves alone", b.ms, "-", b.cyc_valu / it_valu / per_it / NV, b.ghz);
  printf("  %-18s %8.3f %22.2f %26.2f %7.3f\n", "both", c.ms, c.cyc_mfma / it_mfma / 4.0, c.cyc_valu / it_valu / per_it / NV, c.ghz);
  printf("  both / (MFMA alone + VALU alone) = %.3f    both / max = %.3f\n", c.ms / (a.ms + b.ms), c.ms / std::max(a.ms, b.ms));
  printf("  cycles from kernel entry to the end of the role's loop: MFMA waves %.0f alone, %.0f beside; VALU waves %.0f alone, %.0f beside\n",
         a.entry_mfma, c.entry_mfma, b.entry_valu, c.entry_valu);
}

int main() {
  double* d; Stamp* dst;
  (void)hipMalloc(&d, sizeof(double) * 256 * 1024);
  (void)hipMalloc(&dst, sizeof(Stamp) * 256 * 16);
  const int it = 40000;
  // iteration counts chosen so that each role alone takes about the same time (256 cycles per MFMA iteration; a VALU
  // iteration is 8 (17) instructions)
  scenario<1, 0>("v_fma_f64", it, it * 4, d, dst);
  scenario<2, 0>("v_fma_f64", it, it * 3, d, dst);
  scenario<3, 0>("v_fma_f64", it, it * 2, d, dst);
  scenario<3, 1>("fma + add + LDS read", it, it, d, dst);
  scenario<2, 2>("v_add_u32 (32-bit integer VALU)", it, it * 4, d, dst);
  scenario<2, 3>("v_fma_f32", it, it * 4, d, dst);
  return 0;
}
