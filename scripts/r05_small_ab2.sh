# same-box A/B of the small-n evaluators: build_ab/libccgp_base.so (the column's operands requested behind the pivot test, as the
# compiler placed them) against the in-tree library (requested with the pivot, right behind the barrier)
for rep in 1 2; do for L in build_ab/libccgp_base.so ""; do
  echo "== ${L:-in-tree}"
  CCGP_LIB=$L python scripts/logpost_latency.py 2>/dev/null | cut -c1-120
  for W in cfg2 cfg3 cfg5; do CCGP_LIB=$L python bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W', round(d['ms_per_step'],3), 'ms')"; done
done; done
