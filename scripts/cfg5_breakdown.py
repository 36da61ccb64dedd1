"""Kernel time of the Ground-Vibrations prediction tables per train/test pair (size 50: G = 8 grid, one wave per draw;
size 90: 16 x 16 grid, one workgroup per draw)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import ccgp_amd
from ccgp_amd import api
import bench
sets, P = bench.cfg5_inputs()
h = api.Handle(0)
P = np.asfortranarray(P)
for (Xs, ys, Xt) in (sets[0], sets[-1]):
    Xs, Xt = np.asfortranarray(Xs), np.asfortranarray(Xt)
    for _ in range(3):
        h.predict_batch(Xs, ys, 2, P, Xt, 1.0)
    h.enable_timing(True)
    t0 = time.perf_counter(); h.predict_batch(Xs, ys, 2, P, Xt, 1.0); t = time.perf_counter() - t0
    k = h.get_timing()["fused"]
    h.enable_timing(False)
    print("n=%d m=%d: call %.3f ms, kernel %.3f ms (%d launches): %.2f ns per (draw, site)" % (
        Xs.shape[0], Xt.shape[0], 1e3 * t, k[0], k[1], 1e6 * k[0] / (P.shape[0] * Xt.shape[0])))
