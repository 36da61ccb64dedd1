#!/usr/bin/env python3
"""Exploration for the per-point pin against `Size 50 Results 1.txt` (GPU): K = 1 kriging columns at the
recovered mlegp parameters, then the Combined columns over several seeds with sigma2 = mlegp's."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ccgp_amd  # noqa: E402,F401
from ccgp_amd import api, fit  # noqa: E402
from ccgp_amd.rsurface import CombinedGP  # noqa: E402
from ccgp_amd.tables import read_table  # noqa: E402

g = os.path.join(ROOT, "tests", "golden")
rec = json.load(open(os.path.join(g, "gv_mlegp_recovered.json")))
names, res = read_table(os.path.join(g, "data", "gv", "results_50_1.txt"))
col = {n: i for i, n in enumerate(names)}
_, tr = read_table(os.path.join(g, "data", "gv", "train_50_1.txt"))
D, y, Dt, yt = tr[:, :9], tr[:, 9], res[:, :9], res[:, col["y.true"]]
h = api.Handle(0)
theta = np.array(rec["theta"])
row = np.concatenate([[1.0], theta])[None]
mean, var, beta, st = h.predict_batch(D, y, 1, row, Dt, rec["sigma2"])
print("K=1 device mean vs y.hat.single: max abs", np.abs(mean[0] - res[:, col["y.hat.single"]]).max(), "beta", beta[0], rec["beta"])
s2_own, th_own, b_own = fit.ordinary_kriging_sigma2(h, D, y)
ll_m = h.loglik_batch(D, y, 1, row, rec["sigma2"])[0][0]
ll_o = h.loglik_batch(D, y, 1, np.concatenate([[1.0], th_own])[None], s2_own)[0][0]
print("own kriging MLE: sigma2 %.4f loglik %.4f | mlegp: sigma2 %.4f loglik %.4f" % (s2_own, ll_o, rec["sigma2"], ll_m))
print("own theta", th_own)
m_own = h.predict_batch(D, y, 1, np.concatenate([[1.0], th_own])[None], Dt, s2_own)[0][0]
print("own kriging RMSPE %.4f, recorded single RMSPE %.4f, rms(own - recorded) %.4f"
      % (np.sqrt(np.mean((m_own - yt) ** 2)), np.sqrt(np.mean((res[:, col['y.hat.single']] - yt) ** 2)),
         np.sqrt(np.mean((m_own - res[:, col['y.hat.single']]) ** 2))))

gp = CombinedGP("GV", handle=h)
out = {}
for s2name, s2 in (("mlegp", rec["sigma2"]), ("own", s2_own)):
    tabs = []
    for seed in range(6):
        t0 = time.time()
        t = fit.Combined_GP_fit(gp, D, y, Dt, [1.0, 1.0, 0.0], 5000, 1000, 0.5, 20, alpha=0.05, y_new=yt,
                                sigma2=s2, rng=1000 + seed, speculate=4)
        tabs.append(t)
        s = fit.comparison_summary(t)
        print(s2name, "seed", seed, "rmspe %.4f cover %.3f mq %.3f accepted %d/%d  %.1fs  draws mean" % (
            s["rmspe"], s["coverage"], s["mean_quantile"], t["chain"]["accepted"], t["chain"]["proposals"], time.time() - t0),
            t["draws"].mean(axis=0).round(3), "geweke p %.2f" % t["chain"]["geweke_p"])
    Y = np.array([t["y_hat"] for t in tabs])
    LLs = np.array([t["LL"] for t in tabs]); ULs = np.array([t["UL"] for t in tabs]); Q = np.array([t["quant"] for t in tabs])
    yr, llr, ulr, qr = (res[:, col[c]] for c in ("y.hat.Combined", "LL.Combined", "UL.Combined", "Quant.Combined"))
    sd = Y.std(axis=0, ddof=1)
    print(s2name, "rms seed sd of y.hat %.4f | rms(mean_seeds - recorded) %.4f | per-seed rms diff" % (
        np.sqrt((sd ** 2).mean()), np.sqrt(((Y.mean(0) - yr) ** 2).mean())), np.sqrt(((Y - yr) ** 2).mean(axis=1)).round(4))
    print(s2name, "pairwise seed rms diffs", [round(float(np.sqrt(((Y[a] - Y[b]) ** 2).mean())), 4) for a in range(6) for b in range(a + 1, 6)])
    print(s2name, "corr(y.hat, recorded)", [round(float(np.corrcoef(Y[a], yr)[0, 1]), 5) for a in range(6)])
    W = ULs - LLs; wr = ulr - llr
    print(s2name, "interval width: recorded mean %.4f ours per seed" % wr.mean(), W.mean(axis=1).round(4), "rms(width diff)/mean",
          (np.sqrt(((W - wr) ** 2).mean(axis=1)) / wr.mean()).round(4), "seed-to-seed", round(float(np.sqrt(((W[0] - W[1]) ** 2).mean()) / wr.mean()), 4))
    print(s2name, "LL rms diff", np.sqrt(((LLs - llr) ** 2).mean(axis=1)).round(4), "seed-seed", round(float(np.sqrt(((LLs[0] - LLs[1]) ** 2).mean())), 4))
    print(s2name, "Quant: recorded mean %.4f ours" % qr.mean(), Q.mean(axis=1).round(4), "rms diff", np.sqrt(((Q - qr) ** 2).mean(axis=1)).round(4),
          "seed-seed", round(float(np.sqrt(((Q[0] - Q[1]) ** 2).mean())), 4))
h.close()
