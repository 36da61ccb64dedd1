// Probe of the v_mfma_f64_16x16x4_f64 operand / result lane maps with exact integer data
// (asymmetric A and B), printed so that blocked.hip's assumptions can be checked on hardware:
//   A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15], D[i = (lane>>4) + 4r][j = lane&15].
// Build: hipcc --offload-arch=gfx950 -O2 tests/hip/mfma_f64_probe.hip -o gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(const double* A, const double* B, double* D) {
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];      // A is 16x4 row-major
  double b = B[(l >> 4) * 16 + (l & 15)];     // B is 4x16 row-major
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}
int main() {
  double hA[64], hB[64], hD[256], ref[16][16];
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) hA[i * 4 + k] = 1 + i + 100 * k;
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) hB[k * 16 + j] = 3 + 7 * j + (k + 1) * (k + 1);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    double s = 0; for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j]; ref[i][j] = s; }
  double *dA, *dB, *dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  int okA = 1, okB = 1;   // map A: row = (l>>4) + 4r ; map B: row = 4*(l>>4) + r
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    int j = l & 15;
    if (hD[l * 4 + r] != ref[(l >> 4) + 4 * r][j]) okA = 0;
    if (hD[l * 4 + r] != ref[4 * (l >> 4) + r][j]) okB = 0;
  }
  printf("mfma_f64_16x16x4 D map: row=(lane>>4)+4r -> %s ; row=4*(lane>>4)+r -> %s\n", okA ? "MATCH" : "no", okB ? "MATCH" : "no");
  return (okA || okB) ? 0 : 1;
}
