// Is v_mfma_f64_16x16x4_f64 the chain fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0,c)))) bit for bit?  (And in which k order.)
// Build: hipcc --offload-arch=gfx950 -O2 mfma_f64_chain.hip -o mfma_f64_chain ; prints the number of mismatching outputs per order.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

__global__ void k(const double* A, const double* B, const double* C, double* D, int reps) {
  // A: 16 x (4 reps) row-major [i][k], B: (4 reps) x 16 [k][j], C: 16 x 16 [i][j]
  const int lane = threadIdx.x;
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 acc;
  for (int r = 0; r < 4; ++r) acc[r] = C[((lane >> 4) + 4 * r) * 16 + (lane & 15)];
  for (int t = 0; t < reps; ++t) {
    const double a = A[(lane & 15) * 4 * reps + 4 * t + (lane >> 4)];
    const double b = B[(4 * t + (lane >> 4)) * 16 + (lane & 15)];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) D[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[r];
}

int main() {
  const int reps = 2, Kd = 4 * reps;
  int bad_asc = 0, bad_desc = 0, bad_tree = 0, total = 0;
  srand(7);
  for (int trial = 0; trial < 200; ++trial) {
    std::vector<double> A(16 * Kd), B(Kd * 16), C(256), D(256);
    auto rnd = [&](double scale) { return scale * ((double)rand() / RAND_MAX - 0.5) * std::ldexp(1.0, rand() % 12 - 6); };
    for (auto& v : A) v = rnd(1.0);
    for (auto& v : B) v = rnd(1.0);
    for (auto& v : C) v = trial % 2 ? 0.0 : rnd(4.0);
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, 2048); hipMalloc(&dD, 2048);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, reps);
    hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double asc = C[i * 16 + j], desc = C[i * 16 + j], tree = C[i * 16 + j];
        for (int t = 0; t < reps; ++t) {
          for (int q = 0; q < 4; ++q) asc = std::fma(A[i * Kd + 4 * t + q], B[(4 * t + q) * 16 + j], asc);
          for (int q = 3; q >= 0; --q) desc = std::fma(A[i * Kd + 4 * t + q], B[(4 * t + q) * 16 + j], desc);
          double p01 = std::fma(A[i * Kd + 4 * t], B[(4 * t) * 16 + j], A[i * Kd + 4 * t + 1] * B[(4 * t + 1) * 16 + j]);
          double p23 = std::fma(A[i * Kd + 4 * t + 2], B[(4 * t + 2) * 16 + j], A[i * Kd + 4 * t + 3] * B[(4 * t + 3) * 16 + j]);
          tree = tree + (p01 + p23);
        }
        const double got = D[i * 16 + j];
        bad_asc += std::memcmp(&got, &asc, 8) != 0;
        bad_desc += std::memcmp(&got, &desc, 8) != 0;
        bad_tree += std::memcmp(&got, &tree, 8) != 0;
        ++total;
      }
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dD);
  }
  printf("outputs %d: differ from ascending-k fma chain %d, descending %d, pairwise tree %d\n", total, bad_asc, bad_desc, bad_tree);
  return 0;
}
