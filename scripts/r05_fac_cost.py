#!/usr/bin/env python3
"""What keeping the factor costs the likelihood kernel: 1000 draws of the Ground-Vibrations sets (n = 50, 90) through
loglik_batch_dev (plain kernel) and predict_batch_dev (FAC kernel + site kernels), for rocprofv3 --kernel-trace --stats."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ccgp_amd  # noqa
from ccgp_amd import api
import bench
h = api.Handle(0)
sets, P5 = bench.cfg5_inputs()
dev0 = torch.device("cuda", 0)
f64 = dict(dtype=torch.float64, device=dev0)
h.set_stream(torch.cuda.current_stream().cuda_stream)
S = P5.shape[0]
dP = bench.col_major(torch, P5, **f64)
for rep in range(5):
    for (Xs, ys, Xts) in (sets[0], sets[-1]):
        n, m = Xs.shape[0], Xts.shape[0]
        dX, dy = bench.col_major(torch, Xs, **f64), torch.tensor(ys, **f64)
        ll, bt, st = torch.empty(S, **f64), torch.empty(S, **f64), torch.zeros(S, dtype=torch.int32, device=dev0)
        h.loglik_batch_dev(dX, n, 9, dy, 2, dP, S, 1.0, 0, 0.0, ll, bt, st)
        torch.cuda.synchronize()
        mean, var = torch.empty(S * m, **f64), torch.empty(S * m, **f64)
        h.predict_batch_dev(dX, n, 9, dy, 2, dP, S, bench.col_major(torch, Xts, **f64), m, 1.0, mean, var, bt, st)
        torch.cuda.synchronize()
h.close()
