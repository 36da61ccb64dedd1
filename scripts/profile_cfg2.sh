set -e
TAG=${1:-r01x}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG}_cfg2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --workload cfg2 --steps 5 --warmup 2 > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
rocprofv3 --kernel-trace --stats -d $OUT/stats -o t --output-format csv -- python3 $R/bench.py --workload cfg2 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/stats.log 2>&1
tail -c 900 $OUT/bench_cfg2.json
