#!/usr/bin/env python3
"""SQ counters of cov_kernel (the blocked path's matrix build, n = 4096, K = 3, d = 5, 512 matrices per launch) from the three
passes pmc_cov_1..3 of scripts/profile_round4.sh -> profiles/<tag>/pmc_cov_summary.json: instructions per matrix entry
(a wave owns 64 rows x 16 columns of a 64 x 64 tile) and where the wave time goes.  SQ_*_CYCLES / SQ_ACTIVE_INST_* /
SQ_WAIT_* count quad-cycles (MI355X guide, PMC table)."""
import collections, csv, json, sys
tag = sys.argv[1]
tot = collections.defaultdict(float)
launches = 0
for i in (1, 2, 3):
    seen = set()
    for r in csv.DictReader(open('gpurun_out/%s/pmc_cov_%d/t_counter_collection.csv' % (tag, i))):
        if 'cov_kernel' not in r['Kernel_Name']:
            continue
        tot[r['Counter_Name']] += float(r['Counter_Value'])
        seen.add(r['Dispatch_Id'])
    launches = max(launches, len(seen))
waves = tot['SQ_WAVES']
entries = waves * 64 * 16          # every wave of a lower 64 x 64 tile computes 64 rows x 16 columns
out = {"source": "rocprofv3 --pmc (three passes) --kernel-trace -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary; "
                 "cov_kernel<0, scalar columns>, lower 64 x 64 tiles of 512 matrices of order 4096 per launch (K = 3, d = 5)",
       "launches": launches, "raw": dict(tot), "waves": waves, "entries": entries}
if waves:
    per = lambda k: 64.0 * tot[k] / entries      # wave-instructions per 64 entries = instructions per entry per lane
    out["per_entry"] = {
        "valu_instructions": per('SQ_INSTS_VALU'), "salu_instructions": per('SQ_INSTS_SALU'), "lds_instructions": per('SQ_INSTS_LDS'),
        "fp64_fma": per('SQ_INSTS_VALU_FMA_F64'), "fp64_mul": per('SQ_INSTS_VALU_MUL_F64'), "fp64_add": per('SQ_INSTS_VALU_ADD_F64'),
        "fp64_trans": per('SQ_INSTS_VALU_TRANS_F64'), "int32": per('SQ_INSTS_VALU_INT32'), "cvt": per('SQ_INSTS_VALU_CVT')}
    wc = tot['SQ_WAVE_CYCLES']
    if wc:
        out["wave_cycles_per_64_entries"] = 4.0 * wc * 64.0 / entries
        out["fractions_of_wave_time"] = {"valu_issue_active": tot['SQ_ACTIVE_INST_VALU'] / wc, "lds_active": tot['SQ_ACTIVE_INST_LDS'] / wc,
                                         "scalar_active": tot['SQ_ACTIVE_INST_SCA'] / wc,
                                         "waiting_any": tot['SQ_WAIT_ANY'] / wc, "waiting_on_instruction": tot['SQ_WAIT_INST_ANY'] / wc,
                                         "waiting_on_lds": tot['SQ_WAIT_INST_LDS'] / wc}
    gui = tot['GRBM_GUI_ACTIVE'] / 8.0
    if gui and tot['SQ_BUSY_CYCLES']:
        out["active_valu_cycles_over_chip_cycles"] = 4.0 * tot['SQ_ACTIVE_INST_VALU'] / (gui * 1024)   # quad-cycles -> cycles, per SIMD
json.dump(out, open('profiles/%s/pmc_cov_summary.json' % tag, 'w'), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != 'raw'}, indent=1))
