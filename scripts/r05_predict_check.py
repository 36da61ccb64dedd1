#!/usr/bin/env python3
"""Round 5: prediction tables through the kept-factor scheme (one factorisation per draw, lane = test site) against the
extra-row scheme of rounds 2 - 4 (CCGP_OPT_PREDICT_FACTOR 1 / 0): same bits wanted; and the time of both."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ccgp_amd  # noqa
from ccgp_amd import api
import bench


def main():
    h = api.Handle(0)
    rng = np.random.default_rng(3)
    ok = True
    for (n, d, K, S, m) in ((8, 1, 2, 5, 3), (20, 3, 1, 30, 70), (50, 9, 2, 200, 150), (64, 4, 2, 100, 14), (65, 2, 3, 40, 129),
                            (90, 9, 2, 200, 110), (100, 2, 2, 64, 200), (104, 5, 3, 33, 65)):
        X = rng.uniform(size=(n, d))
        y = np.sin(2 * np.pi * X).sum(axis=1)
        P = np.empty((S, K + K * d))
        for b in range(S):
            w = 0.15 + 0.55 * rng.dirichlet(np.ones(K))
            th = np.exp(rng.uniform(np.log(0.5), np.log(30.0), size=(K, d)))
            th[K - 1] = np.maximum(th[K - 1], 25.0)
            P[b] = np.concatenate([w, th.ravel()])
        P[S // 2, K:] = 1e-5     # one draw that cannot be factorised
        Xt = rng.uniform(size=(m, d))
        res = {}
        for opt in (0, 1):
            h.set_option(api.OPT_PREDICT_FACTOR, opt)
            res[opt] = h.predict_batch(X, y, K, P, Xt, 1.7)
        same = all(np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True) for a, b in zip(res[0], res[1]))
        dev = max(np.nanmax(np.abs(np.asarray(a, float) - np.asarray(b, float)) / (np.abs(np.asarray(a, float)) + 1e-300)) for a, b in zip(res[0][:2], res[1][:2]))
        print("n=%d d=%d K=%d S=%d m=%d: identical %s (max rel dev %.2e), failed draws %d / %d" % (
            n, d, K, S, m, same, dev, int((np.asarray(res[0][3]) != 0).sum()), int((np.asarray(res[1][3]) != 0).sum())), flush=True)
        ok = ok and same
    # cfg5 timing, both schemes
    sets, P5 = bench.cfg5_inputs()
    dev0 = torch.device("cuda", 0)
    f64 = dict(dtype=torch.float64, device=dev0)
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    S = P5.shape[0]
    dP = bench.col_major(torch, P5, **f64)
    ds = []
    for (Xs, ys, Xts) in sets:
        ds.append((bench.col_major(torch, Xs, **f64), torch.tensor(ys, **f64), bench.col_major(torch, Xts, **f64), Xs.shape[0], Xts.shape[0],
                   torch.empty(S * Xts.shape[0], **f64), torch.empty(S * Xts.shape[0], **f64), torch.empty(S, **f64),
                   torch.zeros(S, dtype=torch.int32, device=dev0)))
    for opt in (0, 1, 0, 1):
        h.set_option(api.OPT_PREDICT_FACTOR, opt)
        def run():
            for (dX, dy, dXt, n, m, mean, var, bt, st) in ds:
                h.predict_batch_dev(dX, n, 9, dy, 2, dP, S, dXt, m, 1.0, mean, var, bt, st)
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        print("cfg5, 17 sets, scheme %d: %.3f ms per pass" % (opt, 1e3 * (time.perf_counter() - t0) / 20), flush=True)
    h.close()
    print("ALL OK" if ok else "MISMATCH")


if __name__ == "__main__":
    main()
