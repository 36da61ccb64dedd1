# same-box A/B of the small-n evaluators: build_ab/base.so against the in-tree library
for rep in 1 2; do for L in build_ab/base.so ""; do
  echo "== ${L:-in-tree}"
  CCGP_LIB=$L python scripts/logpost_latency.py 2>/dev/null | cut -c1-50
  for W in cfg2 cfg3 cfg5; do CCGP_LIB=$L python bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W', round(d['ms_per_step'],3), 'ms')"; done
done; done
