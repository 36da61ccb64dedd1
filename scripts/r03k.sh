# A/B on one box: update kernel with s_setprio around its MFMA groups (build_ab/libccgp_prio{1,2}.so) against the default
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r03k}
mkdir -p $OUT
cd $R
for v in default prio1 prio2 default prio1 prio2; do
  if [ $v = default ]; then unset CCGP_LIB; else export CCGP_LIB=$R/build_ab/libccgp_$v.so; fi
  timeout -k 10 300 python3 bench.py --steps 5 --no-cpu-baseline --no-secondary > $OUT/b512_$v.json 2> $OUT/b512_$v.err || { tail -5 $OUT/b512_$v.err; exit 1; }
  timeout -k 10 300 python3 bench.py --evals-total 64 --steps 10 --no-cpu-baseline --no-secondary > $OUT/b64_$v.json 2> $OUT/b64_$v.err || { tail -5 $OUT/b64_$v.err; exit 1; }
  python3 -c "
import json
for f in ('b512','b64'):
    r=json.loads(open('$OUT/'+f+'_$v.json').read().strip().splitlines()[-1])
    print(f, '$v', round(r['ms_per_step'],2), 'ms', {k: round(x,2) for k,x in r['kernel_ms_per_step'].items()}, 'digest', r['config']['matches_cpu_potrf_digest'])"
done
