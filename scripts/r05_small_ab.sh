# same-box A/B of the small-n evaluators: build_ab/libccgp_la0.so (-DCCGP_SMALL_LOOKAHEAD=0: the column order of rounds 2 - 4)
# against the in-tree library (column k + 1 published before the bulk of column k's update)
for rep in 1 2; do for L in build_ab/libccgp_la0.so ""; do
  echo "== ${L:-in-tree}"
  CCGP_LIB=$L python scripts/logpost_latency.py 2>/dev/null | cut -c1-120
  for W in cfg2 cfg3 cfg5; do CCGP_LIB=$L python bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W', round(d['ms_per_step'],3), 'ms')"; done
done; done
