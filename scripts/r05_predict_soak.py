#!/usr/bin/env python3
"""Round 5: randomised soak of the kept-factor prediction scheme (n <= 104, K <= 3) against the extra-row scheme of rounds 2 - 4
(CCGP_OPT_PREDICT_FACTOR 1 / 0): random n, d, K, draws, test sites, some draws that cannot be factorised, now and then a small
workspace limit so that the draws go through in several chunks -- every table entry, beta and status must be identical.
usage: python scripts/r05_predict_soak.py [SECONDS] [SEED]"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import ccgp_amd  # noqa: F401
from ccgp_amd import api


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 777)
    h = api.Handle(0)
    t0 = tlast = time.time()
    runs = fails = chunked = failing = 0
    while time.time() - t0 < seconds:
        n = int(rng.integers(2, 105))
        d = int(rng.integers(1, 10))
        K = int(rng.integers(1, 4))
        S = int(rng.integers(1, 1500)) if rng.random() < 0.3 else int(rng.integers(1, 120))
        m = int(rng.integers(1, 400)) if rng.random() < 0.3 else int(rng.integers(1, 70))
        X = rng.uniform(size=(n, d))
        y = np.sin(2 * np.pi * X).sum(axis=1) + 0.05 * rng.normal(size=n)
        P = np.empty((S, K + K * d))
        for b in range(S):
            w = 0.15 + 0.55 * rng.dirichlet(np.ones(K))
            th = np.exp(rng.uniform(np.log(0.5), np.log(30.0), size=(K, d)))
            th[K - 1] = np.maximum(th[K - 1], 25.0)
            P[b] = np.concatenate([w, th.ravel()])
        if rng.random() < 0.3:
            P[rng.integers(0, S), K:] = 1e-5
            failing += 1
        Xt = rng.uniform(size=(m, d))
        small_ws = rng.random() < 0.2
        res = {}
        for opt in (0, 1):
            if small_ws and opt == 1:
                # a fresh handle whose workspace has never grown: the 8 MB limit then decides the chunk (at least 64 draws)
                with api.Handle(0) as h2:
                    h2.set_workspace_limit(8 << 20)
                    res[opt] = h2.predict_batch(X, y, K, P, Xt, 1.7)
                continue
            h.set_option(api.OPT_PREDICT_FACTOR, opt)
            res[opt] = h.predict_batch(X, y, K, P, Xt, 1.7)
        chunked += small_ws
        runs += 1
        if not all(np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True) for a, b in zip(res[0], res[1])):
            fails += 1
            print("DIFFERENT: n=%d d=%d K=%d S=%d m=%d small_ws=%s" % (n, d, K, S, m, small_ws), flush=True)
            break
        if time.time() - tlast > 20:
            tlast = time.time()
            print("%.0f s: %d comparisons (%d with a draw that fails, %d under an 8 MB workspace), all identical; last n=%d d=%d K=%d S=%d m=%d"
                  % (tlast - t0, runs, failing, chunked, n, d, K, S, m), flush=True)
    h.set_option(api.OPT_PREDICT_FACTOR, 1)
    h.close()
    print("SOAK %s: %d comparisons in %.0f s (%d with a failing draw, %d chunked)" % ("FAILED" if fails else "OK", runs, time.time() - t0, failing, chunked), flush=True)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
