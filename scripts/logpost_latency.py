"""us per sequential ccgp_logpost call (Metro's caller, HX:505-512), Qian n = 64 and Ground-Vibrations n = 90, with and without R.Inv.
usage: python scripts/logpost_latency.py"""
import sys, time, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import bench
from ccgp_amd import api

h = api.Handle(0)
X2, y2, P2, K2, s22 = bench.cfg2_inputs()
sets, _ = bench.cfg5_inputs()
gvX, gvy, _ = sets[-1]
for name, X, y, prior in (("Qian n=64", X2, y2, api.PRIOR_INVGAMMA), ("GV n=%d" % gvX.shape[0], gvX, gvy, api.PRIOR_GV)):
    pp = [2.0, 1.0, 2.0, 30.0] if prior == api.PRIOR_INVGAMMA else None
    for want in (True, False):
        t = [math.log(0.5), math.log(20.0), 0.6]
        ref = h.logpost(X, y, 10.0, prior, t, pp, want)
        for _ in range(50):
            h.logpost(X, y, 10.0, prior, t, pp, want)
        t0 = time.perf_counter()
        for i in range(400):
            r = h.logpost(X, y, 10.0, prior, t, pp, want)
        us = (time.perf_counter() - t0) / 400 * 1e6
        print("%-12s R.Inv %-5s %7.1f us per call   val %.12g  sum(R.Inv) %s" % (name, want, us, r["val"], None if not want else "%.12g" % r["R_inv"].sum()), flush=True)
