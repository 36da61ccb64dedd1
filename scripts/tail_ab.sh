# same-box A/B of the tail-strip policies: quarter / half strips (default) against half strips only (rounds 2 - 3)
for rep in 1 2; do for o in "" "--half-tail-strips"; do
python bench.py --evals-total 64 --steps 10 --no-cpu-baseline --no-secondary $o 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slice ${o:-default}', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()}, d['config']['matches_cpu_potrf_digest'])"
done; done
for o in "" "--half-tail-strips"; do for E in 16 128 200; do
python bench.py --evals-total $E --steps 5 --no-cpu-baseline --no-secondary $o 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$E ${o:-default}', round(d['ms_per_step'],2), d['config']['matches_cpu_potrf_digest'])"
done; done
python bench.py --steps 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=512 default', round(d['ms_per_step'],2), d['config']['matches_cpu_potrf_digest'])"
