# counters + kernel stats of the fused evaluator on the Heat-Exchanger grid only (see profile_round2.sh)
set -e
TAG=${1:-r02x}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
C2="python3 $R/bench.py --workload cfg2 --steps 2 --warmup 1 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_cfg2 -o t --output-format csv -- $C2 > $OUT/stats_cfg2.log 2>&1
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_cfg2_$i -o t --output-format csv -- python3 $R/bench.py --workload cfg2 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_cfg2_$i.log 2>&1
done
