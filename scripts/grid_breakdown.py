"""Where a host-pointer ccgp_grid_marginal call (choose.hyperpars, HX:584-595: 624 rows x 1000 Halton nodes, Qian n = 64)
spends its time: launch groups by HIP events, the rest = PCIe + host-side set-up."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import ccgp_amd
from ccgp_amd import api
from ccgp_amd.tables import read_table
h = api.Handle(0)
_, tr = read_table('tests/golden/data/qian_train.txt')
X, y = np.asfortranarray(tr[:, :4]), tr[:, 4].copy()
_, H = read_table('tests/golden/data/hx_hyperpars_matrix.txt')
s2 = float(np.var(y, ddof=1))
for _ in range(3):
    h.grid_marginal(X, y, s2, H, 1000, 50.0, True)
h.enable_timing(True)
t0 = time.perf_counter(); h.grid_marginal(X, y, s2, H, 1000, 50.0, True); t = time.perf_counter() - t0
k = h.get_timing()
h.enable_timing(False)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); h.grid_marginal(X, y, s2, H, 1000, 50.0, True); ts.append(time.perf_counter() - t0)
print("call %.3f ms (untimed median %.3f); launch groups: %s" % (1e3 * t, 1e3 * sorted(ts)[2], {a: round(b[0], 3) for a, b in k.items() if b[1]}))
