#!/bin/bash
# EXPERIMENT: the update launches with one workgroup per CU (dynamic LDS padded to 96 KB) against the usual two
set -e
for rep in 1 2; do
  for lds in 0 98304; do
    echo "== CCGP_X_UPDATE_LDS=$lds B=512"
    CCGP_X_UPDATE_LDS=$lds python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --sched 0 | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['kernel_ms_per_step'], j['roofline']['frac'])"
    echo "== CCGP_X_UPDATE_LDS=$lds B=64"
    CCGP_X_UPDATE_LDS=$lds python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --sched 0 --evals-total 64 | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['kernel_ms_per_step'], j['roofline']['frac'])"
  done
done
