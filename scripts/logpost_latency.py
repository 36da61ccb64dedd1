import sys, time, math
sys.path.insert(0, '.')
import numpy as np
import ccgp_amd
from ccgp_amd import api
from ccgp_amd.tables import read_table
_, tr = read_table('tests/golden/data/qian_train.txt')
X, y = tr[:, :4], tr[:, 4]
s2 = float(np.var(y, ddof=1))
h = api.Handle(0)
theta_t = np.array([math.log(0.3), math.log(15.0), math.log(4.0)])
pars = np.array([7.0, 3.0, 3.0, 28.0])
for want in (True, False):
    h.logpost(X, y, s2, 0, theta_t, pars, want_Rinv=want)
    h.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(100):
        h.logpost(X, y, s2, 0, theta_t, pars, want_Rinv=want)
    el = (time.perf_counter() - t0) / 100
    tm = h.get_timing()
    h.enable_timing(False)
    print('Rinv' if want else 'value', 'us/call %.1f' % (1e6 * el), {k: (round(1e3 * v[0] / max(v[1], 1), 1), v[1]) for k, v in tm.items() if v[1]})
# larger n
for n in (100, 300):
    rng = np.random.default_rng(0)
    Xn = rng.random((n, 3)); yn = np.sin(Xn).sum(axis=1)
    for want in (True, False):
        h.logpost(Xn, yn, 1.0, 2, theta_t, None, want_Rinv=want)
        t0 = time.perf_counter()
        for _ in range(20):
            h.logpost(Xn, yn, 1.0, 2, theta_t, None, want_Rinv=want)
        print(n, 'Rinv' if want else 'value', 'us/call %.1f' % (1e6 * (time.perf_counter() - t0) / 20))
# a speculative Metropolis batch: 7 candidates per ccgp_loglik_batch call (host pointers)
P7 = np.array([np.concatenate([[0.8, 0.2], np.full(4, 0.3 + 0.01 * i), np.full(4, 15.0)]) for i in range(7)])
h.loglik_batch(X, y, 2, P7, s2)
t0 = time.perf_counter()
for _ in range(200):
    h.loglik_batch(X, y, 2, P7, s2)
print('loglik_batch, 7 draws per call: us/call %.1f' % (1e6 * (time.perf_counter() - t0) / 200))
