#!/usr/bin/env python3
"""Counters of the prediction kernels of rounds 2 - 4 (extra-row scheme) on BASELINE config 5 (scripts/r05_cfg5_counters.sh, passes
pmc_cfg5_1..3, collected BEFORE the kept-factor scheme became the default) -> profiles/<tag>/pmc_cfg5_summary.json, per kernel
instantiation: instructions per wave and busy / wait fractions of wave time."""
import collections, csv, json, sys
src, tag = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for i in (1, 2, 3):
    for r in csv.DictReader(open('gpurun_out/%s/pmc_cfg5_%d/t_counter_collection.csv' % (src, i))):
        k = r['Kernel_Name']
        if 'small_reg_kernel' not in k:
            continue
        name = 'small_reg_kernel<16,6,4> (n = 90: 4 waves per (draw, chunk of 62 sites))' if '<16, 6, 4' in k else \
               'small_reg_kernel<8,7,4> (n = 50: 1 wave per (draw, chunk of 30 sites))'
        tot[name][r['Counter_Name']] += float(r['Counter_Value'])
out = {"source": "rocprofv3 --pmc (three passes) --kernel-trace -- python3 bench.py --workload cfg5 --steps 1 --warmup 1 --no-cpu-baseline, "
                 "extra-row scheme (CCGP_OPT_PREDICT_FACTOR 0: the only scheme before round 5)", "kernels": {}}
for name, t in tot.items():
    w, wc = t['SQ_WAVES'], t['SQ_WAVE_CYCLES']
    out["kernels"][name] = {
        "waves": w,
        "per_wave": {"valu_instructions": t['SQ_INSTS_VALU'] / w, "fp64_fma": t['SQ_INSTS_VALU_FMA_F64'] / w, "fp64_mul": t['SQ_INSTS_VALU_MUL_F64'] / w,
                     "fp64_add": t['SQ_INSTS_VALU_ADD_F64'] / w, "int32": t['SQ_INSTS_VALU_INT32'] / w, "salu_instructions": t['SQ_INSTS_SALU'] / w,
                     "lds_instructions": t['SQ_INSTS_LDS'] / w, "wave_cycles": 4.0 * wc / w, "lds_bank_conflict_cycles": t['SQ_LDS_BANK_CONFLICT'] / w},
        "fractions_of_wave_time": {"valu_issue_active": t['SQ_ACTIVE_INST_VALU'] / wc, "lds_active": t['SQ_ACTIVE_INST_LDS'] / wc,
                                   "waiting_any": t['SQ_WAIT_ANY'] / wc, "waiting_on_instruction": t['SQ_WAIT_INST_ANY'] / wc,
                                   "waiting_on_lds": t['SQ_WAIT_INST_LDS'] / wc}}
json.dump(out, open('profiles/%s/pmc_cfg5_summary.json' % tag, 'w'), indent=1)
print(json.dumps(out["kernels"], indent=1))
