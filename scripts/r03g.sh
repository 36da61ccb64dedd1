# A/B on one box of small-n kernel variants: default build against build_ab/libccgp_<variant>.so (variants = arguments 2..)
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r03g}
shift
mkdir -p $OUT
cd $R
for v in default "$@" default "$@"; do
  if [ $v = default ]; then unset CCGP_LIB; else export CCGP_LIB=$R/build_ab/libccgp_$v.so; fi
  for w in cfg2 cfg3 cfg5; do
    timeout -k 10 300 python3 bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > $OUT/${w}_$v.json 2> $OUT/${w}_$v.err || { tail -5 $OUT/${w}_$v.err; exit 1; }
    python3 -c "
import json
r=json.loads(open('$OUT/${w}_$v.json').read().strip().splitlines()[-1])
print('$w', '$v', round(r['ms_per_step'],3), 'ms', 'no-event region', round(r['notiming_ms_per_step'],3))"
  done
done
