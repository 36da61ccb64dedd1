#!/usr/bin/env python3
"""Round 5: where the scheduled sweep's workgroups spend their time (CCGP_OPT_SCHED_POLICY bit 2 -> ccgp_last_sched_profile).
usage: r05_sched_profile.py [evals] [sched] [policy]"""
import sys, os, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ccgp_amd  # noqa
from ccgp_amd import api
import bench


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    sched = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    policy = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    X, y, P, K = bench.cfg4_inputs(B)
    n, d = X.shape
    dev = torch.device("cuda", 0)
    f64 = dict(dtype=torch.float64, device=dev)
    dX, dy, dP = bench.col_major(torch, X, **f64), torch.tensor(y, **f64), bench.col_major(torch, P, **f64)
    ll, bt = torch.empty(B, **f64), torch.empty(B, **f64)
    st = torch.zeros(B, dtype=torch.int32, device=dev)
    h = api.Handle(0)
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    h.set_option(api.OPT_SCHED, sched)
    h.set_option(api.OPT_SCHED_POLICY, policy | 4)
    for _ in range(3):
        h.loglik_batch_dev(dX, n, d, dy, K, dP, B, 1.0, 0, 0.0, ll, bt, st)
    torch.cuda.synchronize()
    h.enable_timing(True)
    h.loglik_batch_dev(dX, n, d, dy, K, dP, B, 1.0, 0, 0.0, ll, bt, st)
    torch.cuda.synchronize()
    tm = h.get_timing()
    pr = h.last_sched_profile()
    names = ["wait", "D", "U", "T", "arrive"]
    out = {"evals": B, "sched": sched, "policy": policy, "sweep_ms": tm["sweep"][0], "workgroups": int(pr.shape[0])}
    for label, sel in (("all", np.ones(len(pr), bool)), ("first_on_cu", pr[:, 7] == 0), ("second_on_cu", pr[:, 7] == 1)):
        if not sel.any():
            continue
        p = pr[sel]
        out[label] = {"workgroups": int(sel.sum()), "tasks_mean": float(p[:, 5].mean()), "tasks_min": float(p[:, 5].min()), "tasks_max": float(p[:, 5].max()),
                      "ms_mean": {k: float(p[:, i].mean()) / 1e3 for i, k in enumerate(names)},
                      "ms_max": {k: float(p[:, i].max()) / 1e3 for i, k in enumerate(names)}}
    out["per_xcd_workgroups"] = [int((pr[:, 6] == q).sum()) for q in range(8)]
    out["per_xcd_tasks"] = [float(pr[pr[:, 6] == q, 5].sum()) for q in range(8)]
    tot = pr[:, 5].sum()
    out["us_per_task"] = {k: float(pr[:, i].sum() / tot) for i, k in enumerate(names)}
    print(json.dumps(out, indent=1))
    h.close()


if __name__ == "__main__":
    main()
