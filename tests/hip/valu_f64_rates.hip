// Issue cost of the fp64 VALU instructions that exp_cov / the covariance kernels are made of, on this chip:
// 8 independent chains per wave, 4 waves per SIMD, no memory traffic.  Prints cycles per wave-instruction
// (a full-rate wave64 fp64 instruction takes 4 cycles on a 16-lane SIMD).  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHAINS 8
#define DEF_KERNEL(NAME, ASM)                                                              \
  __global__ __launch_bounds__(256) void NAME(double* out, int iters, double c) {          \
    double v[CHAINS];                                                                      \
    for (int i = 0; i < CHAINS; ++i) v[i] = 1.0 + 1e-3 * (threadIdx.x + i);                \
    int e = 1;                                                                             \
    for (int it = 0; it < iters; ++it) {                                                   \
      _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) { ASM; }                          \
    }                                                                                      \
    double s = 0;                                                                          \
    for (int i = 0; i < CHAINS; ++i) s += v[i];                                            \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + e;                                    \
  }

DEF_KERNEL(k_fma, asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c)))
DEF_KERNEL(k_mul, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[i]) : "v"(c)))
DEF_KERNEL(k_add, asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[i]) : "v"(c)))
DEF_KERNEL(k_rndne, asm volatile("v_rndne_f64 %0, %0" : "+v"(v[i])))
DEF_KERNEL(k_ldexp, asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(v[i]) : "v"(e)))
DEF_KERNEL(k_cvt, { int t; asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(t) : "v"(v[i])); e += t; })
DEF_KERNEL(k_rcp, asm volatile("v_rcp_f64 %0, %0" : "+v"(v[i])))
DEF_KERNEL(k_addu32, { uint32_t* w = reinterpret_cast<uint32_t*>(&v[i]); asm volatile("v_add_u32 %0, %0, %1" : "+v"(w[1]) : "v"(e)); })
DEF_KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(reinterpret_cast<uint32_t*>(&v[i])[0]) : "v"(e)))

template <class K>
static void run(const char* name, K kern, double* d, int per_iter_extra = 0) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * 4, iters = 20000;   // 4 workgroups of 4 waves per CU: 4 waves per SIMD
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, 10, 1.0000001);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, iters, 1.0000001);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: 4 waves x iters x CHAINS instructions
  const double instr_per_simd = 4.0 * iters * (CHAINS + per_iter_extra);
  printf("%-10s %.3f ms  -> %.2f cycles per wave-instruction at 2.4 GHz\n", name, ms, 2.4e9 * ms * 1e-3 / instr_per_simd);
}

int main() {
  double* d; hipMalloc(&d, sizeof(double) * 256 * 4096);
  run("v_fma_f64", k_fma, d);
  run("v_mul_f64", k_mul, d);
  run("v_add_f64", k_add, d);
  run("v_rndne_f64", k_rndne, d);
  run("v_ldexp_f64", k_ldexp, d);
  run("v_cvt_i32", k_cvt, d, CHAINS);   // plus one v_add_u32 per chain
  run("v_rcp_f64", k_rcp, d);
  run("v_add_u32", k_addu32, d);
  run("v_cndmask", k_cndmask, d);
  return 0;
}
