set -e
mkdir -p gpurun_out/dl
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "blocked_path or n4096 or cutover or non_positive or edge" > gpurun_out/dl/pytest.log 2>&1 || { tail -40 gpurun_out/dl/pytest.log; exit 1; }
tail -2 gpurun_out/dl/pytest.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/dl/b.json 2>gpurun_out/dl/b.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/dl/b.json').read().strip().splitlines()[-1])
print('ms/step %.2f value %.1f'%(d['ms_per_step'],d['value']), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()}, 'TF %.1f'%d['roofline']['achieved'], d['config']['failed_evals'])
PY
