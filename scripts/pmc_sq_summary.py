#!/usr/bin/env python3
"""MFMA-pipe and CU occupancy figures from the SQ pass of scripts/profile_round.sh -> profiles/<tag>/pmc_sq_summary.json"""
import csv, collections, json, re, sys
tag = sys.argv[1]
rows = list(csv.DictReader(open('gpurun_out/%s/pmc_sq/t_counter_collection.csv' % tag)))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    m = re.search(r'(cov_kernel|diag_kernel|chol_trsm\w*|chol_update\w*)', r['Kernel_Name'])
    if not m: continue
    k = 'chol_update' if m.group(1).startswith('chol_update') else m.group(1)
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
out = {}
for k, v in agg.items():
    gui = v['GRBM_GUI_ACTIVE'] / 8.0          # the counter is summed over the 8 XCDs
    out[k] = dict(v)
    out[k]['mfma_busy_frac_of_simd_cycles'] = v['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 256 * 4) if gui else None
    out[k]['cu_busy_frac'] = v['SQ_BUSY_CU_CYCLES'] / (gui * 256) if gui else None
json.dump({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary (sums over all launches of the run)",
           "kernels": out}, open('profiles/%s/pmc_sq_summary.json' % tag, 'w'), indent=1)
print({k: (round(v['mfma_busy_frac_of_simd_cycles'], 3), round(v['cu_busy_frac'], 3)) for k, v in out.items()})
