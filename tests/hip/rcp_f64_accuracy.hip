// How accurate is v_rcp_f64 on this chip, and how many Newton steps does the small-n evaluators' 1 / pivot need?
// Prints the maximum error in ulp of rcp, rcp + 1 step, rcp + 2 steps against a correctly rounded 1 / x (host long
// double), over 2^22 arguments spread over 40 binades.   hipcc --offload-arch=gfx950 -O3 -o rcp_acc rcp_f64_accuracy.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double r = __builtin_amdgcn_rcp(v);
  r0[i] = r;
  r = fma(fma(-v, r, 1.0), r, r);
  r1[i] = r;
  r = fma(fma(-v, r, 1.0), r, r);
  r2[i] = r;
}

int main() {
  const int n = 1 << 22;
  std::vector<double> x(n), a(n), b(n), c(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double m = 1.0 + (double)(s >> 11) / 9007199254740992.0;
    x[i] = std::ldexp(m, (int)(s % 40) - 30);
  }
  double *dx, *d0, *d1, *d2;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
  hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; ++i) {
    const long double t = 1.0L / (long double)x[i];
    const double ulp = std::ldexp(1.0, std::ilogb((double)t) - 52);
    e0 = std::fmax(e0, (double)fabsl((long double)a[i] - t) / ulp);
    e1 = std::fmax(e1, (double)fabsl((long double)b[i] - t) / ulp);
    e2 = std::fmax(e2, (double)fabsl((long double)c[i] - t) / ulp);
  }
  printf("v_rcp_f64 max error: %.3g ulp;  + 1 Newton step: %.3g ulp;  + 2 steps: %.3g ulp\n", e0, e1, e2);
  return 0;
}
