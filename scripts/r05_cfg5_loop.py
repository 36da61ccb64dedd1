#!/usr/bin/env python3
"""cfg5's 17 prediction tables, a few passes, for rocprofv3 --kernel-trace --stats.  usage: r05_cfg5_loop.py [scheme 0|1]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ccgp_amd  # noqa
from ccgp_amd import api
import bench
opt = int(sys.argv[1]) if len(sys.argv) > 1 else 1
h = api.Handle(0)
sets, P5 = bench.cfg5_inputs()
dev0 = torch.device("cuda", 0)
f64 = dict(dtype=torch.float64, device=dev0)
h.set_stream(torch.cuda.current_stream().cuda_stream)
h.set_option(api.OPT_PREDICT_FACTOR, opt)
S = P5.shape[0]
dP = bench.col_major(torch, P5, **f64)
for rep in range(5):
    for (Xs, ys, Xts) in sets:
        n, m = Xs.shape[0], Xts.shape[0]
        mean, var = torch.empty(S * m, **f64), torch.empty(S * m, **f64)
        h.predict_batch_dev(bench.col_major(torch, Xs, **f64), n, 9, torch.tensor(ys, **f64), 2, dP, S, bench.col_major(torch, Xts, **f64), m, 1.0,
                            mean, var, torch.empty(S, **f64), torch.zeros(S, dtype=torch.int32, device=dev0))
    torch.cuda.synchronize()
h.close()
