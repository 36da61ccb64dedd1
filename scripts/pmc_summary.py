#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/profile_round.sh into
profiles/<tag>/pmc_traffic.json (HBM bytes per launch per kernel; gfx950 corrections as in the MI355X guide:
FETCH_SIZE counts 128-B requests at 64 B for 16 B/lane coalesced reads -> doubled; WRITE_SIZE as is;
both counters are in units of 64 B... the csv already reports bytes/ kilobytes as labelled by rocprofv3)."""
import csv, json, re, sys, collections
tag = sys.argv[1]
suffix = sys.argv[2] if len(sys.argv) > 2 else ''          # '' = the 512-matrix run, '64' = the 64-matrix slice, 's1' = --sched 1 (round 5), 'fc' = --fused-cov (round 4)
nmat = int(sys.argv[3]) if len(sys.argv) > 3 else 512
base = 'gpurun_out/%s' % tag
def per_kernel(path, counter):
    agg, launches = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter: continue
        m = re.search(r'(cov_kernel|diag_kernel|chol_trsm\w*|chol_update|chol_sched)', r['Kernel_Name'])
        if not m: continue
        k = 'chol_update' if m.group(1).startswith('chol_update') else ('chol_trsm_kernel' if m.group(1).startswith('chol_trsm') else m.group(1))
        agg[k] += float(r['Counter_Value']); launches[k].add(r['Dispatch_Id'])
    return agg, {k: len(v) for k, v in launches.items()}
f, nf = per_kernel(base + '/pmc_fetch%s/t_counter_collection.csv' % suffix, 'FETCH_SIZE')
w, nw = per_kernel(base + '/pmc_write%s/t_counter_collection.csv' % suffix, 'WRITE_SIZE')
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary%s; MI355X, cfg4 n=4096, %d matrices per launch" % ((" --evals-total %d" % nmat if nmat != 512 else "") + (" --fused-cov" if suffix == "fc" else "") + (" --sched 1" if suffix == "s1" else "") + (" --sched 0" if suffix == "64s0" else ""), nmat),
       "matrices_per_launch": nmat,
       "correction": "counters reported in KiB; gfx950: FETCH_SIZE counts 128-B requests at 64 B for 16 B/lane coalesced reads -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE taken as is",
       "kernels": {}}
for k in f:
    fr = f[k] * 1024.0
    wr = w.get(k, 0.0) * 1024.0
    out["kernels"][k] = {"launches": nf[k], "fetch_bytes_raw": fr, "fetch_bytes_corrected": 2 * fr, "write_bytes": wr,
                         "hbm_bytes_per_launch": (2 * fr + wr) / nf[k]}
json.dump(out, open('profiles/%s/pmc_traffic%s.json' % (tag, suffix), 'w'), indent=1)
print(json.dumps(out["kernels"], indent=1))
