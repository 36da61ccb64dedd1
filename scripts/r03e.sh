# fused panel solve: GPU suite, then A/B of the 512 grid and the 64 slice with and without --no-fuse-trsm (same box)
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r03e}
mkdir -p $OUT
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -2 $OUT/pytest_gpu.log
for v in fused unfused fused unfused; do
  F=""; if [ $v = unfused ]; then F="--no-fuse-trsm"; fi
  timeout -k 10 300 python3 bench.py --steps 5 --no-cpu-baseline --no-secondary $F > $OUT/b512_$v.json 2> $OUT/b512_$v.err || { tail -5 $OUT/b512_$v.err; exit 1; }
  timeout -k 10 300 python3 bench.py --evals-total 64 --steps 10 --no-cpu-baseline --no-secondary $F > $OUT/b64_$v.json 2> $OUT/b64_$v.err || { tail -5 $OUT/b64_$v.err; exit 1; }
  python3 -c "
import json
for f in ('b512','b64'):
    r=json.loads(open('$OUT/'+f+'_$v.json').read().strip().splitlines()[-1])
    print(f, '$v', round(r['ms_per_step'],2), 'ms', {k: round(x,2) for k,x in r['kernel_ms_per_step'].items()}, 'digest', r['config']['matches_cpu_potrf_digest'])"
done
