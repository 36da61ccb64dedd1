#!/usr/bin/env python3
"""Round 5: randomised soak of the dataflow tile scheduler.  For SECONDS (default 480) draw a shape -- n, matrices per chunk,
job (likelihood, prediction rows, gradient) -- and a scheduler variant, run it through the launches and through the scheduler
and compare every output bit for bit.  One progress line per 20 s; stops at the first difference or error.
usage: python scripts/r05_sched_soak.py [SECONDS] [SEED]"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import ccgp_amd  # noqa: F401
from ccgp_amd import api

VARIANTS = [(1, 11), (2, 11), (2, 11), (1, 3), (2, 3), (1, 9), (2, 0), (1, 10)]


def draws(rng, B, K, d):
    P = np.empty((B, K + K * d))
    for b in range(B):
        w = 0.15 + 0.55 * rng.dirichlet(np.ones(K))
        th = np.exp(rng.uniform(np.log(0.5), np.log(50.0), size=(K, d)))
        th[K - 1] = np.maximum(th[K - 1], 20.0)
        P[b] = np.concatenate([w, th.ravel()])
    return P


def same(a, b):
    return all(np.array_equal(np.asarray(x), np.asarray(y), equal_nan=True) for x, y in zip(a, b))


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 480.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
    h = api.Handle(0)
    t0 = tlast = time.time()
    runs = fails = 0
    kinds = {}
    while time.time() - t0 < seconds:
        job = rng.choice(["loglik", "loglik", "loglik", "predict", "grad"])
        if job == "loglik":
            n = int(rng.integers(129, 4200))
            cap = max(2, int(1.2e10 / (8.0 * (n + 128) * n)))            # <= 12 GB of matrices
            B = int(rng.integers(1, min(cap, 160) + 1))
        elif job == "predict":
            n = int(rng.integers(129, 2300))
            B = int(rng.integers(1, 24))
        else:
            n = int(rng.integers(129, 1100))
            B = int(rng.integers(1, 6))
        d, K = int(rng.integers(1, 6)), int(rng.integers(1, 4))
        X = rng.uniform(size=(n, d))
        y = np.sin(2 * np.pi * X).sum(axis=1) + 0.1 * rng.normal(size=n)
        P = draws(rng, B, K, d)
        if rng.random() < 0.15:
            P[rng.integers(0, B), K:] = 1e-4                               # one evaluation that fails
        if job == "loglik":
            mode, tau2 = ((0, 0.0), (1, 3.0))[int(rng.integers(0, 2))]
            fn = lambda: h.loglik_batch(X, y, K, P, 1.3, mode, tau2)
        elif job == "predict":
            Xt = rng.uniform(size=(int(rng.integers(1, 300)), d))
            fn = lambda: h.predict_batch(X, y, K, P, Xt, 2.0)
        else:
            fn = lambda: h.loglik_grad_batch(X, y, K, P, 1.0)
        h.set_option(api.OPT_SCHED, 0)
        ref = fn()
        sched, policy = VARIANTS[int(rng.integers(0, len(VARIANTS)))]
        h.set_option(api.OPT_SCHED, sched)
        h.set_option(api.OPT_SCHED_POLICY, policy)
        got = fn()
        h.set_option(api.OPT_SCHED, 3)
        h.set_option(api.OPT_SCHED_POLICY, 11)
        runs += 1
        kinds[job] = kinds.get(job, 0) + 1
        if not same(ref, got):
            fails += 1
            print("DIFFERENT: job %s n=%d d=%d K=%d B=%d sched=%d policy=%d" % (job, n, d, K, B, sched, policy), flush=True)
            break
        if time.time() - tlast > 20:
            tlast = time.time()
            print("%.0f s: %d comparisons (%s), all identical; last: %s n=%d B=%d sched=%d policy=%d" % (
                tlast - t0, runs, ", ".join("%s %d" % kv for kv in sorted(kinds.items())), job, n, B, sched, policy), flush=True)
    h.close()
    print("SOAK %s: %d comparisons in %.0f s (%s)" % ("FAILED" if fails else "OK", runs, time.time() - t0,
                                                       ", ".join("%s %d" % kv for kv in sorted(kinds.items()))), flush=True)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
