"""Experiment: the 64-matrix slice (one GPU's share of the 512-point grid at N = 8) as ONE chunk on one stream against
2 / 4 sub-chunks on separate streams (separate handles), so that the partial last round of one chunk's launch overlaps the
next launch of the other chain."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import ccgp_amd
from ccgp_amd import api
import bench
B = 64
X, y, P, K = bench.cfg4_inputs(B)
n, d = X.shape
dev = torch.device('cuda', 0)
f64 = dict(dtype=torch.float64, device=dev)
dX, dy = bench.col_major(torch, X, **f64), torch.tensor(y, **f64)
def make(parts):
    hs = []
    for lo in range(0, B, B // parts):
        hi = lo + B // parts
        h = api.Handle(0)
        s = torch.cuda.Stream()
        h.set_stream(s.cuda_stream)
        dP = bench.col_major(torch, P[lo:hi], **f64)
        out = (torch.empty(hi - lo, **f64), torch.empty(hi - lo, **f64), torch.zeros(hi - lo, dtype=torch.int32, device=dev))
        h.reserve(n, d, K, hi - lo, 0)
        hs.append((h, s, dP, out, hi - lo))
    return hs
ref = None
for parts in (1, 2, 4, 1, 2, 4):
    hs = make(parts)
    def step():
        for (h, s, dP, out, nb) in hs:
            h.loglik_batch_dev(dX, n, d, dy, K, dP, nb, 1.0, 0, 0.0, *out)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 10
    ll = torch.cat([o[3][0] for o in hs]).cpu().numpy()
    if ref is None:
        ref = ll
    print("%d chunk(s) on %d stream(s): %.3f ms per 64 matrices; same bits as one chunk: %s" % (parts, parts, 1e3 * t, np.array_equal(ll, ref)))
    for (h, *_r) in hs:
        h.close()
