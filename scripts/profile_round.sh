# Round profile set: default bench line, rocprofv3 kernel stats, PMC traffic passes (separate runs).
# usage (GPU box): bash scripts/profile_round.sh r01h
set -e
TAG=${1:-r01x}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 600 $OUT/bench_default.json
B="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o t --output-format csv -- $B > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o t --output-format csv -- $B > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o t --output-format csv -- $B > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace -d $OUT/pmc_sq -o t --output-format csv -- $B > $OUT/pmc_sq.log 2>&1
ls -R $OUT | head -40
