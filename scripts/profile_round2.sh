# Round profile set (GPU box): default bench line, rocprofv3 kernel stats, PMC traffic passes (separate runs, as the
# MI355X guide prescribes), SQ counters; the same for the Heat-Exchanger grid (cfg2) and for the 64-matrix slice one
# GPU gets at N = 8.   usage: bash scripts/profile_round2.sh r02x
set -e
TAG=${1:-r02x}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python3 $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 400 $OUT/bench_default.json; echo
python3 $R/bench.py --evals-total 64 --steps 10 --no-cpu-baseline --no-secondary > $OUT/bench_slice64.json 2> $OUT/bench_slice64.err
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
# Heat-Exchanger grid (fused small-n evaluator)
C2="python3 $R/bench.py --workload cfg2 --steps 2 --warmup 1 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_cfg2 -o t --output-format csv -- $C2 > $OUT/stats_cfg2.log 2>&1
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_cfg2_$i -o t --output-format csv -- python3 $R/bench.py --workload cfg2 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_cfg2_$i.log 2>&1
done
cd $R
ls $OUT | head -40
