#!/usr/bin/env python3
"""Round 5: the dataflow scheduler against the launch-per-phase sweep, bit for bit, on shapes that exercise every task kind
(plain likelihood, prediction rows, identity rows of inverse / gradient), before anything is timed."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ccgp_amd  # noqa
from ccgp_amd import api


VARIANTS = [(0, 11), (1, 11), (2, 11), (1, 3), (1, 9), (2, 0)]


def draws(B, K, d, rng):
    P = np.empty((B, K + K * d))
    for b in range(B):
        w = 0.15 + 0.55 * rng.dirichlet(np.ones(K))
        th = np.exp(rng.uniform(np.log(0.5), np.log(50.0), size=(K, d)))
        th[K - 1] = np.maximum(th[K - 1], 20.0)
        P[b] = np.concatenate([w, th.ravel()])
    return P


def main():
    rng = np.random.default_rng(5)
    ok = True
    h = api.Handle(0)
    for (n, d, K, B, m) in ((300, 3, 2, 20, 0), (257, 2, 2, 3, 40), (1000, 5, 3, 37, 200), (2100, 4, 2, 9, 130), (4096, 5, 3, 16, 0)):
        X = rng.uniform(size=(n, d))
        y = np.sin(2 * np.pi * X).sum(axis=1)
        P = draws(B, K, d, rng)
        Xt = rng.uniform(size=(max(m, 1), d))
        res = {}
        # (sched, policy): policy 11 = default (backlog rule + XCD-local synchronisation + chaining); 3 = no chaining;
        # 9 = agent-scope fences and stealing, chaining; 0 = agent-scope, no backlog rule, no chaining
        for sched in VARIANTS:
            h.set_option(api.OPT_SCHED, sched[0])
            h.set_option(api.OPT_SCHED_POLICY, sched[1])
            t0 = time.perf_counter()
            ll, beta, st = h.loglik_batch(X, y, K, P, 1.0)
            out = [ll, beta, st]
            if m:
                out += list(h.predict_batch(X, y, K, P, Xt, 1.0)[:2])
            if n <= 1000:
                out += list(h.loglik_grad_batch(X, y, K, P[:5], 1.0))
            res[sched] = out
            print("n=%d B=%d m=%d sched=%s: %.1f ms  ll[0]=%.12g bad=%d" % (n, B, m, sched, 1e3 * (time.perf_counter() - t0), ll[0], int((st != 0).sum())), flush=True)
        for sched in VARIANTS[1:]:
            same = all(np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True) for a, b in zip(res[VARIANTS[0]], res[sched]))
            print("   sched=%s bit-identical to launches: %s" % (sched, same), flush=True)
            ok = ok and same
    h.close()
    print("ALL OK" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
