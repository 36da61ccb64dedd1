# A/B on one box of blocked-path variants: default build against build_ab/libccgp_<variant>.so (variants = arguments 2..):
# the 64-matrix slice, per-kernel milliseconds
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r03v}
shift
mkdir -p $OUT
cd $R
for v in default "$@" default "$@"; do
  if [ $v = default ]; then unset CCGP_LIB; else export CCGP_LIB=$R/build_ab/libccgp_$v.so; fi
  timeout -k 10 300 python3 bench.py --evals-total 64 --steps 10 --no-cpu-baseline --no-secondary > $OUT/b64_$v.json 2> $OUT/b64_$v.err || { tail -5 $OUT/b64_$v.err; exit 1; }
  python3 -c "
import json
r=json.loads(open('$OUT/b64_$v.json').read().strip().splitlines()[-1])
print('$v', round(r['ms_per_step'],2), 'ms', {k: round(x,2) for k,x in r['kernel_ms_per_step'].items()}, 'digest', r['config']['matches_cpu_potrf_digest'])"
done
