"""Where a ccgp_loglik_grad_batch call (Qian n = 64, 65 536 draws, host pointers) spends its time: kernel (HIP events
around the fused launch), the rest = PCIe + pinned staging + the ctypes wrapper."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import ccgp_amd
from ccgp_amd import api
from ccgp_amd.tables import read_table
h = api.Handle(0)
_, tr = read_table('tests/golden/data/qian_train.txt')
X, y = np.asfortranarray(tr[:, :4]), tr[:, 4].copy()
rng = np.random.default_rng(0)
B = 65536
P = np.asfortranarray(np.array([np.concatenate([[p, 1 - p], np.full(4, t1), np.full(4, t2)]) for p, t1, t2 in
                                zip(rng.uniform(0.5, 0.9, B), rng.uniform(0.2, 1.0, B), rng.uniform(5, 30, B))]))
for name, fn in (("grad", lambda: h.loglik_grad_batch(X, y, 2, P, 37.0)), ("loglik", lambda: h.loglik_batch(X, y, 2, P, 37.0))):
    fn(); fn()
    h.enable_timing(True)
    t0 = time.perf_counter(); fn(); t = time.perf_counter() - t0
    k = h.get_timing()["fused"]
    h.enable_timing(False)
    t0 = time.perf_counter(); fn(); t2 = time.perf_counter() - t0
    print("%s: call %.3f ms (untimed %.3f), kernel %.3f ms in %d launch(es)" % (name, 1e3 * t, 1e3 * t2, k[0], k[1]))
