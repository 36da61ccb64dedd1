#!/usr/bin/env python3
"""Reference values for the benchmark's own workload (BASELINE config 4: n = 4096, d = 5, K = 3, 512 draws), so that
bench.py can check what it just timed instead of only `isfinite`.

Every draw: log-likelihood and beta by potrf + two forward solves on the CPU (the recipe of
tests/test_gpu_parity.py::test_n4096_against_lapack_and_invariances).  All 512 go through the compiled CPU
evaluator (oracle/cpu_baseline: OpenBLAS dpotrf / dtrsv); 6 of them are re-derived here with scipy.linalg on the
oracle's covariance matrix and must agree to 1e-10, which pins the compiled evaluator for this workload.
Run from the repo root (CPU, ~2 min on 8 cores):  python tests/golden/make_cfg4_digest.py"""
import json
import math
import os
import sys

import numpy as np
import scipy.linalg as sla

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import ccgp_oracle as orc  # noqa: E402
from oracle.cpu_baseline import loader as cpu  # noqa: E402


def main():
    X, y, P, K = bench.cfg4_inputs(512)
    n, d = X.shape
    ll, beta, st = cpu.loglik_batch(X, y, K, P, 1.0, 0, 0.0)
    assert not st.any()
    for b in (0, 1, 63, 64, 300, 511):
        w, Th = orc.unpack_params(P[b], K, d)
        L = sla.cholesky(orc.mixed_corr_matrix_general(X, w, Th), lower=True)
        zy = sla.solve_triangular(L, y, lower=True)
        z1 = sla.solve_triangular(L, np.ones(n), lower=True)
        b0 = (z1 @ zy) / (z1 @ z1)
        c = float(np.sum(w ** 2))
        want = -0.5 * (n * math.log(2 * math.pi) + n * math.log(c) + 2 * np.log(np.diag(L)).sum()
                       + np.sum((zy - b0 * z1) ** 2) / c)
        print(b, ll[b], want, abs(ll[b] - want) / abs(want), beta[b], b0)
        assert abs(ll[b] - want) <= 1e-10 * abs(want) and abs(beta[b] - b0) <= 1e-8 * max(abs(b0), 1e-3)
    out = dict(workload="bench.cfg4_inputs(512): n=4096, d=5, K=3, sigma2=1, profile-beta likelihood",
               loglik=[float(v) for v in ll], beta=[float(v) for v in beta])
    with open(os.path.join(HERE, "cfg4_loglik_512.json"), "w") as fh:
        json.dump(out, fh)
    print("written", len(out["loglik"]))


if __name__ == "__main__":
    main()
