// Practical ceiling of v_mfma_f64_16x16x4_f64 on this chip: every SIMD issues back-to-back
// MFMAs on register operands (no memory traffic), 16 independent accumulators per wave.
// Prints TFLOP/s for 1 and 2 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 2) void peak(double* out, int iters) {
  d4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  double* d; hipMalloc(&d, sizeof(double) * 256 * 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
    int grid = 256 * wgs_per_cu, iters = 20000;
    hipLaunchKernelGGL(peak, dim3(grid), dim3(256), 0, 0, d, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(peak, dim3(grid), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * 16 * 2048.0;
    printf("waves/SIMD=%d: %.2f ms, %.1f TFLOP/s, implied cycles/MFMA at 2.4 GHz = %.1f\n", wgs_per_cu, ms,
           flops / ms / 1e9, 2.4e9 * (ms * 1e-3) / ((double)iters * 16 * wgs_per_cu));
  }
  return 0;
}
