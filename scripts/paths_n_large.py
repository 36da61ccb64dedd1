"""The blocked-path entry points other than the plain likelihood, once each at a size where kernels dominate, for
`rocprofv3 --kernel-trace --stats` (which kernel takes the time?): prediction refactorising and from a kept factor set
(n = 4096, 16 draws, 128 sites), logpost with its explicit inverse (n = 1000 and 4096), the log-determinants of candidate designs
(n = 1000).  usage: python scripts/paths_n_large.py"""
import sys, time, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import bench
from ccgp_amd import api

X, y, P, K = bench.cfg4_inputs(16)
h = api.Handle(0)
Xt = bench.cfg4_predict_sites(X.shape[1])


def timed(name, fn, reps=2):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    print("%-60s %9.2f ms" % (name, (time.perf_counter() - t0) / reps * 1e3), flush=True)
    return out


timed("predict_batch n=4096, 16 draws, 128 sites (refactorising)", lambda: h.predict_batch(X, y, K, P, Xt, 1.0))
fs = timed("factor_batch n=4096, 16 draws", lambda: h.factor_batch(X, y, K, P, 1.0), reps=1)
timed("factor set -> predict 128 sites", lambda: fs.predict(Xt))
fs.free()
for n in (1000, 4096):
    theta_t = [math.log(3.0), math.log(40.0), 0.8]
    timed("logpost (GV script, isotropic) with R.Inv, n=%d" % n,
          lambda: h.logpost(X[:n], y[:n], 1.0, api.PRIOR_GV, theta_t, None, True))
    timed("logpost value only, n=%d" % n, lambda: h.logpost(X[:n], y[:n], 1.0, api.PRIOR_GV, theta_t, None, False))
