set -e
mkdir -p gpurun_out/d16
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "blocked_path or n4096 or cutover or non_positive or edge" > gpurun_out/d16/pytest.log 2>&1 || { tail -40 gpurun_out/d16/pytest.log; exit 1; }
tail -2 gpurun_out/d16/pytest.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
for O in 0 1; do
  CCGP_DIAG_OLD=$O timeout -k 10 200 $B > gpurun_out/d16/old$O.json 2>gpurun_out/d16/old$O.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/d16/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, 'ms/step %.2f'%d['ms_per_step'], {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()}, 'TF %.1f'%d['roofline']['achieved'], d['config']['failed_evals'])
    except Exception as e: print(f,'ERR',e)
PY
