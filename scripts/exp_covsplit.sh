set -e
mkdir -p gpurun_out/cs
CCGP_COV_SPLIT=4 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "blocked_path or n4096 or n8192 or chunked" > gpurun_out/cs/pytest.log 2>&1 || { tail -40 gpurun_out/cs/pytest.log; exit 1; }
tail -2 gpurun_out/cs/pytest.log
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary"
for C in 0 2 4 6 8 12; do
  CCGP_COV_SPLIT=$C timeout -k 10 200 $B > gpurun_out/cs/c$C.json 2>gpurun_out/cs/c$C.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/cs/*.json'), key=lambda s:int(s.split('/c')[-1].split('.')[0])):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, 'ms/step %.2f'%d['ms_per_step'], {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()}, 'TF %.1f'%d['roofline']['achieved'], d['config']['failed_evals'])
    except Exception as e: print(f,'ERR',e)
PY
