// Issue cost of the fp64 VALU instructions that exp_cov / the covariance and small-n evaluators are made of, on this
// chip -- and WHICH of (clock, operand ports, the probe itself) sets it.  Round 3's version of this probe timed a launch
// with HIP events and divided by an ASSUMED 2.4 GHz: 5.65 - 5.78 "cycles" per wave-instruction for v_fma / v_mul /
// v_add_f64 where a 16-lane SIMD needs 4.  This version measures cycles and clock IN the kernel
// (MI355X_MICROARCH.md, DVFS give-back item 6): every wave stamps s_memtime (shader cycles) and s_memrealtime
// (100 MHz) around its loop, so
//     cycles per wave-instruction = d(s_memtime) * waves_per_SIMD / instructions issued by the SIMD's waves
//     clock                       = d(s_memtime) / d(s_memrealtime) * 100 MHz
// and it varies what the round-3 review asked for: 1 / 2 / 4 / 8 waves per SIMD, VOP3 v_fma_f64 vs VOP2 v_fmac_f64,
// VGPR vs SGPR vs inline-constant sources, repeated vs distinct source registers, 8 vs 16 independent chains.
// No memory traffic inside the loop.  Build: hipcc --offload-arch=gfx950 -O3 valu_f64_rates.hip -o valu_rates
// Run plain for the table; under `rocprofv3 --pmc GRBM_GUI_ACTIVE` (etc., one counter set per pass) with argument
// `pmc` it launches only the long v_fma_f64 / v_add_u32 kernels, so the counter rows are easy to read.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Stamp { unsigned long long t0, t1, rt; unsigned hwid, xcc; };

// CH independent chains, the body is one instruction per chain and iteration
#define DEF_KERNEL(NAME, CH, PRE, ASM)                                                     \
  __global__ __launch_bounds__(256) void NAME(double* out, Stamp* st, int iters, double c, double c2) { \
    double v[CH], a[CH], b[CH];                                                            \
    for (int i = 0; i < CH; ++i) {                                                         \
      v[i] = 1.0 + 1e-3 * (threadIdx.x + i);                                               \
      a[i] = 1.0 + 1e-9 * (threadIdx.x + 3 * i);                                           \
      b[i] = 1e-12 * (i + 1);                                                              \
    }                                                                                      \
    int e = 1;                                                                             \
    PRE;                                                                                   \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
    for (int it = 0; it < iters; ++it) {                                                   \
      _Pragma("unroll") for (int i = 0; i < CH; ++i) { ASM; }                              \
    }                                                                                      \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
    double s = 0;                                                                          \
    for (int i = 0; i < CH; ++i) s += v[i] + a[i] + b[i];                                  \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + e;                                    \
    if ((threadIdx.x & 63) == 0) {                                                         \
      Stamp& s_ = st[blockIdx.x * 4 + (threadIdx.x >> 6)];                                 \
      s_.t0 = t0; s_.t1 = t1; s_.rt = r1 - r0;                                             \
      s_.hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);   /* HW_ID: simd [5:4] cu [11:8] sh [12] se [15:13] */ \
      s_.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   /* XCC_ID */                  \
    }                                                                                      \
  }

// fp64, three sources
DEF_KERNEL(k_fma_rep, 8, , asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c)))                    // round 3's form: src1 == src2
DEF_KERNEL(k_fma_dist, 8, , asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(v[i]) : "v"(a[i]), "v"(b[i])))     // three distinct VGPR pairs
DEF_KERNEL(k_fma_dist16, 16, , asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(v[i]) : "v"(a[i]), "v"(b[i])))
DEF_KERNEL(k_fmac, 8, , asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(v[i]) : "v"(a[i]), "v"(b[i])))            // VOP2 encoding of the same operation
DEF_KERNEL(k_fma_sgpr, 8, , asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[i]) : "s"(c), "v"(b[i])))        // one source from an SGPR pair
DEF_KERNEL(k_fma_inl, 8, , asm volatile("v_fma_f64 %0, %0, 1.0, %1" : "+v"(v[i]) : "v"(b[i])))                // inline constant
DEF_KERNEL(k_fma_2src, 8, , asm volatile("v_fma_f64 %0, %0, 1.0, 0.5" : "+v"(v[i])))                          // ONE register source
// full-entropy mantissas in every source (the clock the chip holds depends on the data: MI355X_MICROARCH.md, DVFS give-back)
#define RANDOMISE for (int i = 0; i < 8; ++i) { unsigned long long z = (threadIdx.x * 8 + i + 1) * 0x9E3779B97F4A7C15ull; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32; \
    a[i] = __longlong_as_double(0x3ff0000000000000ull | (z >> 12)); z *= 0x94D049BB133111EBull; z ^= z >> 31; b[i] = (__longlong_as_double(0x3ff0000000000000ull | (z >> 12)) - 1.5) * 1e-3; v[i] = a[i] * 0.7; }
DEF_KERNEL(k_fma_rand, 8, RANDOMISE, asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(v[i]) : "v"(a[i]), "v"(b[i])))
// fp64, two sources / one source
DEF_KERNEL(k_mul, 8, , asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[i]) : "v"(c)))
DEF_KERNEL(k_add, 8, , asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[i]) : "v"(b[i])))
DEF_KERNEL(k_add_inl, 8, , asm volatile("v_add_f64 %0, %0, 1.0" : "+v"(v[i])))
DEF_KERNEL(k_rndne, 8, , asm volatile("v_rndne_f64 %0, %0" : "+v"(v[i])))
DEF_KERNEL(k_ldexp, 8, , asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(v[i]) : "v"(e)))
// 32-bit references
DEF_KERNEL(k_fma_f32, 8, , { float* w = reinterpret_cast<float*>(&v[i]); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(w[0]) : "v"(w[1])); })
DEF_KERNEL(k_pk_fma_f32, 8, , asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(a[i])))
DEF_KERNEL(k_addu32, 8, , { uint32_t* w = reinterpret_cast<uint32_t*>(&v[i]); asm volatile("v_add_u32 %0, %0, %1" : "+v"(w[1]) : "v"(e)); })
DEF_KERNEL(k_mov64, 8, , asm volatile("v_mov_b64 %0, %1" : "=v"(v[i]) : "v"(a[i])))
// the mix exp_cov is made of: 8 fma with one SGPR source + 1 add + 1 max + ldexp, per "entry"
DEF_KERNEL(k_expmix, 8, , {
  asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[i]) : "s"(c), "v"(b[i]));
  asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
})

using Kern = void (*)(double*, Stamp*, int, double, double);

struct Count { int waves, simds; double cyc_per_inst; };
struct Result { double ghz, ms; int nsimd; std::vector<Count> by_count; };

static int run(Kern kern, int waves_per_simd, int chains, int per_chain, int iters, double* d, Stamp* dst, Result* res) {
  // one workgroup = 4 waves = one wave per SIMD of its CU; 256 w workgroups give w waves per SIMD only ON AVERAGE
  const int grid = 256 * waves_per_simd;
  hipEvent_t e0, e1;
  HIPCHECK(hipEventCreate(&e0));
  HIPCHECK(hipEventCreate(&e1));
  // clocks settle under THIS instruction mix: untimed launches for ~0.3 s, then the measured one
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, dst, iters, 1.0000001, 0.5);
  HIPCHECK(hipDeviceSynchronize());
  HIPCHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, dst, iters, 1.0000001, 0.5);
  HIPCHECK(hipEventRecord(e1));
  HIPCHECK(hipEventSynchronize(e1));
  float ms1 = 0;
  HIPCHECK(hipEventElapsedTime(&ms1, e0, e1));
  const int reps = std::max(1, (int)(300.0 / std::max(ms1, 0.01f)));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, dst, iters, 1.0000001, 0.5);
  HIPCHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, dst, iters, 1.0000001, 0.5);
  HIPCHECK(hipEventRecord(e1));
  HIPCHECK(hipEventSynchronize(e1));
  float ms = 0;
  HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<Stamp> st((size_t)grid * 4);
  HIPCHECK(hipMemcpy(st.data(), dst, sizeof(Stamp) * st.size(), hipMemcpyDeviceToHost));
  // WHERE the waves ran decides how many share a SIMD (the dispatcher does not spread 256 w workgroups evenly: round 4's
  // first version of this probe assumed it did and read 2 - 3.75 "cycles" per fp64 instruction).  Group the stamps by
  // (XCC, SE, SH, CU, SIMD) from the hardware-id registers; per SIMD: instructions of its waves / (last end - first start)
  std::map<unsigned long long, std::vector<const Stamp*>> simds;
  std::vector<double> ghz;
  for (const Stamp& q : st) {
    const unsigned long long key = ((unsigned long long)(q.xcc & 0xf) << 32) | (q.hwid & 0xfff0u & ~0xc0u);   // drop wave_id [3:0], pipe_id [7:6]
    simds[key].push_back(&q);
    if (q.rt) ghz.push_back((double)(q.t1 - q.t0) / (double)q.rt * 0.1);
  }
  std::sort(ghz.begin(), ghz.end());
  const double inst_per_wave = (double)iters * chains * per_chain;
  std::map<int, std::vector<double>> by_count;   // waves on the SIMD -> cycles per wave-instruction
  for (auto& kv : simds) {
    unsigned long long lo = ~0ull, hi = 0;
    for (const Stamp* q : kv.second) { lo = std::min(lo, q->t0); hi = std::max(hi, q->t1); }
    by_count[(int)kv.second.size()].push_back((double)(hi - lo) / (inst_per_wave * kv.second.size()));
  }
  res->ghz = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];
  res->ms = ms;
  res->nsimd = (int)simds.size();
  res->by_count.clear();
  for (auto& kv : by_count) {
    std::sort(kv.second.begin(), kv.second.end());
    res->by_count.push_back({kv.first, (int)kv.second.size(), kv.second[kv.second.size() / 2]});
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return 0;
}

int main(int argc, char** argv) {
  const bool pmc = argc > 1 && !strcmp(argv[1], "pmc");
  double* d;
  Stamp* dst;
  HIPCHECK(hipMalloc(&d, sizeof(double) * 256 * 8 * 256));
  HIPCHECK(hipMalloc(&dst, sizeof(Stamp) * 256 * 8 * 4));
  struct Row { const char* name; Kern k; int chains, per_chain; const char* what; };
  const Row rows[] = {
      {"v_fma_f64 v,v,c,c", k_fma_rep, 8, 1, "VOP3, src1 == src2 (round 3's probe)"},
      {"v_fma_f64 v,a,b,v", k_fma_dist, 8, 1, "VOP3, three distinct VGPR pairs"},
      {"v_fma_f64 x16", k_fma_dist16, 16, 1, "same, 16 chains"},
      {"v_fma_f64 random", k_fma_rand, 8, 1, "three distinct sources, full-entropy mantissas"},
      {"v_fmac_f64 v,a,b", k_fmac, 8, 1, "VOP2 encoding"},
      {"v_fma_f64 v,v,s,b", k_fma_sgpr, 8, 1, "one SGPR-pair source"},
      {"v_fma_f64 v,v,1.0,b", k_fma_inl, 8, 1, "one inline constant"},
      {"v_fma_f64 v,v,1.0,0.5", k_fma_2src, 8, 1, "ONE register source"},
      {"v_mul_f64", k_mul, 8, 1, ""},
      {"v_add_f64 v,v,b", k_add, 8, 1, ""},
      {"v_add_f64 v,v,1.0", k_add_inl, 8, 1, "one register source"},
      {"v_rndne_f64", k_rndne, 8, 1, ""},
      {"v_ldexp_f64", k_ldexp, 8, 1, ""},
      {"v_fma_f32", k_fma_f32, 8, 1, "32-bit reference"},
      {"v_pk_fma_f32", k_pk_fma_f32, 8, 1, "64-bit operands, fp32 lanes"},
      {"v_add_u32", k_addu32, 8, 1, "32-bit integer reference"},
      {"v_mov_b64", k_mov64, 8, 1, "64-bit move"},
      {"fma(s) + add pair", k_expmix, 8, 2, "two fp64 instructions per chain"},
  };
  const int iters = 20000;
  if (pmc) {
    // counter passes: one long launch set per kernel, nothing else
    Result r;
    printf("pmc mode: v_fma_f64 (distinct) then v_add_u32, 4 waves per SIMD, 5 launches each\n");
    for (int q = 0; q < 5; ++q) hipLaunchKernelGGL(k_fma_dist, dim3(1024), dim3(256), 0, 0, d, dst, iters * 5, 1.0000001, 0.5);
    HIPCHECK(hipDeviceSynchronize());
    for (int q = 0; q < 5; ++q) hipLaunchKernelGGL(k_addu32, dim3(1024), dim3(256), 0, 0, d, dst, iters * 5, 1.0000001, 0.5);
    HIPCHECK(hipDeviceSynchronize());
    (void)r;
    return 0;
  }
  printf("cycles per wave-instruction of a SIMD, by the number of waves that actually shared it (SIMDs in that class), clock in the kernel\n");
  printf("%-24s %6s %6s  %s\n", "instruction", "grid/256", "GHz", "waves on the SIMD: cycles per instruction (SIMDs)");
  for (const Row& row : rows) {
    for (int w : {1, 2, 4, 6, 8}) {
      Result r{};
      if (run(row.k, w, row.chains, row.per_chain, iters, d, dst, &r)) return 1;
      // the whole launch as one number: SIMD-cycles of the launch per wave-instruction issued (includes ramp and tail)
      const double launch_cpi = r.ms * 1e-3 * r.ghz * 1e9 * 1024.0 / (256.0 * w * 4 * (double)iters * row.chains * row.per_chain);
      printf("%-24s %6d %6.3f  launch %.2f |", row.name, w, r.ghz, launch_cpi);
      for (const Count& c : r.by_count) printf(" %d: %.2f (%d)", c.waves, c.cyc_per_inst, c.simds);
      printf("   %s\n", row.what);
    }
  }
  return 0;
}
