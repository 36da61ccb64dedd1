"""Throughput of the analytic gradient (ccgp_loglik_grad_batch, the north_star's "+ gradient") next to the plain
likelihood: Qian n = 64 (fused path), maximin-100, and the blocked path at n = 1024 / 4096."""
import sys, time, math
sys.path.insert(0, '.')
import numpy as np
import ccgp_amd
from ccgp_amd import api
from ccgp_amd.tables import read_table
import bench
h = api.Handle(0)
_, tr = read_table('tests/golden/data/qian_train.txt')
X, y = tr[:, :4], tr[:, 4]
rng = np.random.default_rng(0)
for B in (1, 64, 4096, 65536):
    P = np.array([np.concatenate([[p, 1 - p], np.full(4, t1), np.full(4, t2)]) for p, t1, t2 in
                  zip(rng.uniform(0.5, 0.9, B), rng.uniform(0.2, 1.0, B), rng.uniform(5, 30, B))])
    h.loglik_grad_batch(X, y, 2, P, 37.0)
    t0 = time.perf_counter(); h.loglik_grad_batch(X, y, 2, P, 37.0); tg = time.perf_counter() - t0
    h.loglik_batch(X, y, 2, P, 37.0)
    t0 = time.perf_counter(); h.loglik_batch(X, y, 2, P, 37.0); tl = time.perf_counter() - t0
    print('n=64 B=%d: grad %.3f ms (%.0f /s), loglik %.3f ms (%.0f /s), ratio %.1f' % (B, 1e3 * tg, B / tg, 1e3 * tl, B / tl, tg / tl))
for n, B in ((1024, 64), (4096, 16)):
    Xc, yc, Pc, K = bench.cfg4_inputs(B, n=n)
    h.loglik_grad_batch(Xc, yc, K, Pc, 1.0)
    t0 = time.perf_counter(); h.loglik_grad_batch(Xc, yc, K, Pc, 1.0); tg = time.perf_counter() - t0
    h.loglik_batch(Xc, yc, K, Pc, 1.0)
    t0 = time.perf_counter(); h.loglik_batch(Xc, yc, K, Pc, 1.0); tl = time.perf_counter() - t0
    print('n=%d B=%d: grad %.2f ms (%.1f /s), loglik %.2f ms (%.1f /s), ratio %.1f' % (n, B, 1e3 * tg, B / tg, 1e3 * tl, B / tl, tg / tl))
