set -e
mkdir -p gpurun_out/dt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "blocked_path or n4096 or cutover or non_positive or edge" > gpurun_out/dt/pytest.log 2>&1 || { tail -30 gpurun_out/dt/pytest.log; exit 1; }
tail -2 gpurun_out/dt/pytest.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
for S in 0 1 2; do
  CCGP_STRIPS=$S timeout -k 10 200 $B > gpurun_out/dt/s$S.json 2>gpurun_out/dt/s$S.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/dt/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, 'ms/step %.2f'%d['ms_per_step'], {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()}, 'TF %.1f'%d['roofline']['achieved'], d['config']['failed_evals'])
    except Exception as e: print(f,'ERR',e)
PY
