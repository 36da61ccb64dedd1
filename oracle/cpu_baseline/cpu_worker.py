#!/usr/bin/env python3
"""One worker PROCESS of bench.py's CPU sweep (concurrent evaluations x threads per evaluation): evaluates its draws one
at a time with `inner` threads (OpenMP for the covariance columns, OpenBLAS for dpotrf / dtrsv).  Test / bench
infrastructure.  Protocol on stdin / stdout: prints "ready" after loading and one warm-up evaluation, waits for a line, runs,
prints "done <seconds> <sum of log-likelihoods>".
usage: cpu_worker.py <inputs.npz> <first draw> <draws> <inner threads>"""
import os
import sys
import time

inner = int(sys.argv[4])
os.environ["OMP_NUM_THREADS"] = str(inner)
os.environ["OPENBLAS_NUM_THREADS"] = str(inner)

import numpy as np  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle.cpu_baseline import loader as cpu  # noqa: E402


def main():
    z = np.load(sys.argv[1])
    lo, cnt = int(sys.argv[2]), int(sys.argv[3])
    X, y, P, K = z["X"], z["y"], z["P"], int(z["K"])
    sigma2, mode, tau2 = float(z["sigma2"]), int(z["mode"]), float(z["tau2"])
    rows = P[[(lo + i) % P.shape[0] for i in range(cnt)]]
    cpu.loglik_seq(X, y, K, rows[:1], sigma2, mode, tau2, inner)      # warm-up: thread pools, page faults
    print("ready", flush=True)
    sys.stdin.readline()
    t0 = time.perf_counter()
    ll, _, _ = cpu.loglik_seq(X, y, K, rows, sigma2, mode, tau2, inner)
    print("done %.6f %.17g" % (time.perf_counter() - t0, float(np.nansum(ll))), flush=True)


if __name__ == "__main__":
    main()
