# A/B on one box: small-n evaluators with the polynomial exp (default build) against the table exp (build_ab/libccgp_tab.so)
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r03c}
mkdir -p $OUT
cd $R
for v in poly tab poly tab; do
  if [ $v = tab ]; then export CCGP_LIB=$R/build_ab/libccgp_tab.so; else unset CCGP_LIB; fi
  for w in cfg2 cfg3 cfg5; do
    timeout -k 10 300 python3 bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > $OUT/${w}_$v.json 2> $OUT/${w}_$v.err || { tail -5 $OUT/${w}_$v.err; exit 1; }
    python3 -c "
import json
r=json.loads(open('$OUT/${w}_$v.json').read().strip().splitlines()[-1])
print('$w', '$v', round(r['ms_per_step'],3), 'ms', 'kernel', round(r['roofline']['avg_launch_ms'],3))"
  done
done
unset CCGP_LIB
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -2 $OUT/pytest_gpu.log
