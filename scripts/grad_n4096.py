"""Blocked-path gradient at BASELINE config 4's size against the plain likelihood: ms per call for B draws (host pointers) and the
launch groups' device time (HIP events).  usage: python scripts/grad_n4096.py [B]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (before libccgp: see INTEGRATION.md)
import bench
from ccgp_amd import api

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
X, y, P, K = bench.cfg4_inputs(512)
sigma2 = 1.0
h = api.Handle(0)
P = np.ascontiguousarray(P[:B])
for name, fn in (("loglik", lambda: h.loglik_batch(X, y, K, P, sigma2)), ("loglik + gradient", lambda: h.loglik_grad_batch(X, y, K, P, sigma2))):
    fn()
    t0 = time.perf_counter()
    for _ in range(3):
        out = fn()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    h.enable_timing(True)
    fn()
    tm = {k: round(v[0], 2) for k, v in h.get_timing().items() if v[1]}
    h.enable_timing(False)
    print("%-18s B = %d: %8.2f ms per call  (%.2f ms per draw)  device groups %s  failed %d" % (name, B, ms, ms / B, tm, int(np.count_nonzero(out[-1]))))
