# small-n micro-optimisation check: GPU suite + cfg2/cfg3/cfg5 timings
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r03f}
mkdir -p $OUT
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -2 $OUT/pytest_gpu.log
for rep in 1 2; do
for w in cfg2 cfg3 cfg5; do
  timeout -k 10 300 python3 bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > $OUT/${w}_$rep.json 2> $OUT/${w}_$rep.err || { tail -5 $OUT/${w}_$rep.err; exit 1; }
  python3 -c "
import json
r=json.loads(open('$OUT/${w}_$rep.json').read().strip().splitlines()[-1])
print('$w', round(r['ms_per_step'],3), 'ms')"
done
done
