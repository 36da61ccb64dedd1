# Round-5 profile set (GPU box), all from the HEAD binary.  usage: bash scripts/profile_round5.sh r05
#   default bench line (512-point grid: launch-per-phase sweep, the library's choice at that size), with cpu_baseline + secondary
#   the 64-matrix slice (one GPU's share at N = 8: the dataflow scheduler, the library's choice there) and the same with --sched 0
#   the 512-point grid with the scheduler forced (--sched 1)
#   rocprofv3 kernel stats + the two PMC traffic passes for each of the three
#   the scheduler's per-workgroup time accounts (slice and full grid, one and two workgroups per CU)
#   cfg2 / cfg3 / cfg5 bench lines, kernel stats of cfg5
set -e
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export CCGP_SCHED_TIMEOUT_MS=10000
python3 $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 400 $OUT/bench_default.json; echo
N="--no-cpu-baseline --no-secondary"
python3 $R/bench.py --evals-total 64 --steps 10 $N > $OUT/bench_slice64.json 2> $OUT/bench_slice64.err
python3 $R/bench.py --evals-total 64 --steps 10 $N --sched 0 > $OUT/bench_slice64_sched0.json 2> $OUT/bench_slice64_sched0.err
python3 $R/bench.py --steps 5 $N --sched 1 > $OUT/bench_sched1.json 2> $OUT/bench_sched1.err
python3 $R/bench.py --evals-total 64 --steps 10 $N > $OUT/bench_slice64_b.json 2> /dev/null
python3 $R/bench.py --evals-total 64 --steps 10 $N --sched 0 > $OUT/bench_slice64_sched0_b.json 2> /dev/null
python3 $R/bench.py --workload cfg2 --steps 10 --warmup 2 > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
python3 $R/bench.py --workload cfg3 --steps 10 --warmup 2 > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
python3 $R/bench.py --workload cfg5 --steps 10 --warmup 2 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
for a in "64 2 11" "64 1 11" "512 1 11" "512 2 11"; do set -- $a; python3 $R/scripts/r05_sched_profile.py $1 $2 $3 > $OUT/sched_account_$1_sched$2.json 2> /dev/null; done
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o t --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 $N > $OUT/stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o t --output-format csv -- $B > $OUT/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o t --output-format csv -- $B > $OUT/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace -d $OUT/pmc_sq -o t --output-format csv -- $B > $OUT/pmc_sq.log 2>&1
# scheduler forced on the full grid
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/statss1 -o t --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 $N --sched 1 > $OUT/statss1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetchs1 -o t --output-format csv -- $B --sched 1 > $OUT/pmc_fetchs1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_writes1 -o t --output-format csv -- $B --sched 1 > $OUT/pmc_writes1.log 2>&1
# the 64-matrix slice: scheduler (default there) and launches
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats64 -o t --output-format csv -- $B --evals-total 64 --steps 2 > $OUT/stats64.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch64 -o t --output-format csv -- $B --evals-total 64 > $OUT/pmc_fetch64.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write64 -o t --output-format csv -- $B --evals-total 64 > $OUT/pmc_write64.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats64s0 -o t --output-format csv -- $B --evals-total 64 --steps 2 --sched 0 > $OUT/stats64s0.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch64s0 -o t --output-format csv -- $B --evals-total 64 --sched 0 > $OUT/pmc_fetch64s0.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write64s0 -o t --output-format csv -- $B --evals-total 64 --sched 0 > $OUT/pmc_write64s0.log 2>&1
# cfg5: the three kernels of the kept-factor prediction
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_cfg5 -o t --output-format csv -- python3 $R/bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/stats_cfg5.log 2>&1
for W in cfg2 cfg3; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_$W -o t --output-format csv -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $OUT/stats_$W.log 2>&1
done
cd $R
ls $OUT | head -80
