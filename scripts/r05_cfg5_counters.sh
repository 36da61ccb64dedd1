# cfg5 (Ground-Vibrations prediction tables): bench line, kernel stats and the three SQ passes cfg2 / cfg3 have
set -e
TAG=${1:-r05cfg5}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python3 $R/bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_cfg5 -o t --output-format csv -- python3 $R/bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/stats_cfg5.log 2>&1
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_cfg5_$i -o t --output-format csv -- python3 $R/bench.py --workload cfg5 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_cfg5_$i.log 2>&1
done
cd $R
head -c 600 $OUT/bench_cfg5.json; echo
grep small_reg $OUT/stats_cfg5/*kernel_stats.csv | head -5
