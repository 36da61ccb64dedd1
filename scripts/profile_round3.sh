# Round-3 profile set (GPU box): default bench line, rocprofv3 kernel stats, PMC traffic passes (separate runs, as the
# MI355X guide prescribes), SQ counters; the same for the 64-matrix slice one GPU gets at N = 8, for the Heat-Exchanger
# grid (cfg2) and the 2-D grid (cfg3).   usage: bash scripts/profile_round3.sh r03
set -e
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python3 $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 300 $OUT/bench_default.json; echo
python3 $R/bench.py --evals-total 64 --steps 10 --no-cpu-baseline --no-secondary > $OUT/bench_slice64.json 2> $OUT/bench_slice64.err
python3 $R/bench.py --workload cfg2 --steps 10 --warmup 2 > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
python3 $R/bench.py --workload cfg3 --steps 10 --warmup 2 > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
python3 $R/bench.py --workload cfg5 --steps 10 --warmup 2 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o t --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o t --output-format csv -- $B > $OUT/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o t --output-format csv -- $B > $OUT/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace -d $OUT/pmc_sq -o t --output-format csv -- $B > $OUT/pmc_sq.log 2>&1
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
cd $R
ls $OUT | head -60
