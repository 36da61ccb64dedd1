#!/usr/bin/env python3
"""Counters of the fused small-n evaluator on the Heat-Exchanger grid (scripts/profile_round2.sh, passes pmc_cfg2_1..3)
-> profiles/<tag>/pmc_cfg2_summary.json: instructions per evaluation (one wave = one 64 x 64 evaluation at G = 8) and
busy fractions.  SQ_*_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles (MI355X guide, PMC table)."""
import collections, csv, json, sys
tag = sys.argv[1]
wl = sys.argv[2] if len(sys.argv) > 2 else 'cfg2'      # cfg2 (n = 64: one wave per evaluation) | cfg3 (n = 100)
# cfg3: one workgroup of 4 waves per evaluation on the 16 x 16 grid (rounds 1 - 3, CCGP_OPT_SMALL_GRID16 / bench.py --small-grid16), ONE wave per
# evaluation on the 8 x 8 grid with 13 x 13 blocks per thread (round 4); pass 4 as third argument for the former
waves_per_eval = int(sys.argv[3]) if len(sys.argv) > 3 else 1
tot = collections.defaultdict(float)
launches = 0
for i in (1, 2, 3):
    seen = set()
    for r in csv.DictReader(open('gpurun_out/%s/pmc_%s_%d/t_counter_collection.csv' % (tag, wl, i))):
        if 'small_reg_kernel' not in r['Kernel_Name']:
            continue
        tot[r['Counter_Name']] += float(r['Counter_Value'])
        seen.add(r['Dispatch_Id'])
    launches = max(launches, len(seen))
waves = tot['SQ_WAVES'] / waves_per_eval     # = evaluations
out = {"source": "rocprofv3 --pmc (three passes) --kernel-trace -- python3 bench.py --workload %s --steps 1 --warmup 1 --no-cpu-baseline; " % wl +
                 ("small_reg_kernel<8,8,1,full>, one wave per evaluation, 624 000 evaluations per launch" if wl == 'cfg2' else
                  "small_reg_kernel<16,7,1>, one workgroup (4 waves) per evaluation, 103 680 evaluations per launch; per-evaluation "
                  "figures are sums over the 4 waves" if waves_per_eval == 4 else
                  "small_reg_kernel<8,13,1>, one wave per evaluation, 103 680 evaluations per launch"),
       "launches": launches, "raw": dict(tot)}
if waves:
    per = lambda k: tot[k] / waves
    out["per_evaluation"] = {
        "valu_instructions": per('SQ_INSTS_VALU'), "salu_instructions": per('SQ_INSTS_SALU'), "lds_instructions": per('SQ_INSTS_LDS'),
        "fp64_fma": per('SQ_INSTS_VALU_FMA_F64'), "fp64_mul": per('SQ_INSTS_VALU_MUL_F64'), "fp64_add": per('SQ_INSTS_VALU_ADD_F64'),
        "fp64_trans": per('SQ_INSTS_VALU_TRANS_F64'), "int32": per('SQ_INSTS_VALU_INT32'), "cvt": per('SQ_INSTS_VALU_CVT'),
        "wave_cycles": 4.0 * per('SQ_WAVE_CYCLES')}
    wc = tot['SQ_WAVE_CYCLES']
    if wc:
        out["fractions_of_wave_time"] = {"valu_issue_active": tot['SQ_ACTIVE_INST_VALU'] / wc, "lds_active": tot['SQ_ACTIVE_INST_LDS'] / wc,
                                         "waiting_any": tot['SQ_WAIT_ANY'] / wc, "waiting_on_instruction": tot['SQ_WAIT_INST_ANY'] / wc,
                                         "waiting_on_lds": tot['SQ_WAIT_INST_LDS'] / wc}
    out["lds_bank_conflict_cycles_per_eval"] = per('SQ_LDS_BANK_CONFLICT')
json.dump(out, open('profiles/%s/pmc_%s_summary.json' % (tag, wl), 'w'), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != 'raw'}, indent=1))
