# Round-4 profile set (GPU box), all from the HEAD binary: default bench line, rocprofv3 kernel stats, PMC traffic passes
# (separate runs, as the MI355X guide prescribes), SQ counters of the update; the 64-matrix slice one GPU gets at N = 8;
# the Heat-Exchanger grid (cfg2) and the 2-D grid (cfg3) with their three SQ passes; NEW: the same three SQ passes for
# cov_kernel, the fp64 VALU cadence probe (in-kernel clock) with a GRBM_GUI_ACTIVE pass beside it, and the MFMA / VALU
# co-issue probe.   usage: bash scripts/profile_round4.sh r04
set -e
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python3 $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 300 $OUT/bench_default.json; echo
python3 $R/bench.py --evals-total 64 --steps 10 --no-cpu-baseline --no-secondary > $OUT/bench_slice64.json 2> $OUT/bench_slice64.err
python3 $R/bench.py --workload cfg2 --steps 10 --warmup 2 > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
python3 $R/bench.py --workload cfg3 --steps 10 --warmup 2 > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
python3 $R/bench.py --workload cfg3 --small-grid16 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg3_grid16.json 2> $OUT/bench_cfg3_grid16.err
python3 $R/bench.py --workload cfg5 --steps 10 --warmup 2 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
$R/tests/hip/valu_rates > $OUT/valu_f64_rates.txt 2>&1
$R/tests/hip/mfma_valu_overlap > $OUT/mfma_valu_overlap.txt 2>&1
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o t --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o t --output-format csv -- $B > $OUT/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o t --output-format csv -- $B > $OUT/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace -d $OUT/pmc_sq -o t --output-format csv -- $B > $OUT/pmc_sq.log 2>&1
# cov_kernel writing every tile (default) against covariance tiles generated in the update (--fused-cov = CCGP_OPT_FUSED_COV):
# alternating bench lines, kernel stats and the two traffic passes of the fused data flow beside the default ones above
: > $OUT/fused_cov_ab.txt
for o in "" "--fused-cov" "" "--fused-cov"; do
  python3 $R/bench.py --steps 5 --no-cpu-baseline --no-secondary $o 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-16s %7.2f ms per step  %s  digest %s' % ('$o' or 'default', d['ms_per_step'], {k: round(v, 2) for k, v in d['kernel_ms_per_step'].items()}, d['config']['matches_cpu_potrf_digest']))" >> $OUT/fused_cov_ab.txt
done
cat $OUT/fused_cov_ab.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/statsfc -o t --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --fused-cov > $OUT/statsfc.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetchfc -o t --output-format csv -- $B --fused-cov > $OUT/pmc_fetchfc.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_writefc -o t --output-format csv -- $B --fused-cov > $OUT/pmc_writefc.log 2>&1
# cov_kernel: the three SQ passes that cfg2 / cfg3 have
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_cov_$i -o t --output-format csv -- $B > $OUT/pmc_cov_$i.log 2>&1
done
# the 64-matrix slice
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats64 -o t --output-format csv -- $B --evals-total 64 --steps 2 > $OUT/stats64.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch64 -o t --output-format csv -- $B --evals-total 64 > $OUT/pmc_fetch64.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write64 -o t --output-format csv -- $B --evals-total 64 > $OUT/pmc_write64.log 2>&1
# the hyperprior grids (fused small-n evaluator): cfg2 = Heat-Exchanger (n = 64), cfg3 = 2-D anisotropic (n = 100)
for W in cfg2 cfg3; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_$W -o t --output-format csv -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $OUT/stats_$W.log 2>&1
  i=0
  for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_${W}_$i -o t --output-format csv -- python3 $R/bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_${W}_$i.log 2>&1
  done
done
# the clock under the fp64 VALU probe, from the counters (GRBM_GUI_ACTIVE / 8 / duration), beside the in-kernel figure
timeout -k 10 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_valu_clock -o t --output-format csv -- $R/tests/hip/valu_rates pmc > $OUT/pmc_valu_clock.log 2>&1
cd $R
ls $OUT | head -80
