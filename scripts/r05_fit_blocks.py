#!/usr/bin/env python3
"""Round 5: the end-to-end fit item of bench.py by itself (sequential Metro and blocks of 4 / 6 / 8 proposals per device call)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import bench
from ccgp_amd import api

h = api.Handle(0)
bench.end_to_end_fit_item(h)          # warm-up (first calls, workspace)
for rep in range(2):
    it = bench.end_to_end_fit_item(h)
    print(json.dumps({k: (v if not isinstance(v, dict) else {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items()})
                      for k, v in it.items() if k not in ("workload", "note")}), flush=True)
h.close()
