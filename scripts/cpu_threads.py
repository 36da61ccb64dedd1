import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from threadpoolctl import threadpool_limits, threadpool_info
import bench
from oracle import ccgp_oracle as orc
X, y, P, K = bench.cfg4_inputs(4)
print(os.cpu_count(), [ (p['internal_api'], p['num_threads']) for p in threadpool_info()])
for nt in (8, 16, 32, 64, 128):
    with threadpool_limits(limits=nt):
        t0 = time.perf_counter()
        w, Th = orc.unpack_params(P[0], K, 5)
        orc.loglik_general(X, y, w, Th, 1.0, 0, 0.0)
        print(nt, 'threads: %.2f s per evaluation' % (time.perf_counter() - t0), flush=True)
