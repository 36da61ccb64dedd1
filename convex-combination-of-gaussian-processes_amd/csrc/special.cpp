// Host entry points of the quadrature helpers (special_math.h) behind ccgp_halton_base2 / ccgp_qigamma.
#include "special_math.h"

#include "ccgp_internal.h"

namespace ccgp {

void halton_base2(int N, double* out) {
  for (int i = 1; i <= N; ++i) out[i - 1] = halton2(static_cast<unsigned>(i));
}

double qigamma_host(double p, double alpha, double beta) { return qigamma(p, alpha, beta); }

}  // namespace ccgp
