# A/B on one box with per-launch tables: kernel trace of one 512-matrix step for the default build and each variant
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r03w}
shift
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
for v in default "$@"; do
  if [ $v = default ]; then unset CCGP_LIB; else export CCGP_LIB=$R/build_ab/libccgp_$v.so; fi
  timeout -k 10 300 python3 bench.py --steps 3 --no-cpu-baseline --no-secondary > $OUT/b512_$v.json 2> $OUT/b512_$v.err || { tail -5 $OUT/b512_$v.err; exit 1; }
  python3 -c "
import json
r=json.loads(open('$OUT/b512_$v.json').read().strip().splitlines()[-1])
print('$v', round(r['ms_per_step'],2), 'ms', {k: round(x,2) for k,x in r['kernel_ms_per_step'].items()}, 'digest', r['config']['matches_cpu_potrf_digest'])"
  timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/trace_$v -o t --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/trace_$v.log 2>&1 || { tail -5 $OUT/trace_$v.log; exit 1; }
  cp $(find $OUT/trace_$v -name 't_kernel_trace.csv' | head -1) $OUT/kernel_trace_$v.csv
  rm -rf $OUT/trace_$v
done
python3 scripts/trace_table.py $(for v in default "$@"; do echo $OUT/kernel_trace_$v.csv; done) > $OUT/table.txt
cat $OUT/table.txt
