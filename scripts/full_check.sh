set -e
mkdir -p gpurun_out/full
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/full/pytest.log 2>&1 || { tail -40 gpurun_out/full/pytest.log; exit 1; }
tail -3 gpurun_out/full/pytest.log
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/full/bench.json').read().strip().splitlines()[-1])
print('ms/step %.2f value %.1f'%(d['ms_per_step'],d['value']), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()}, 'TF %.1f'%d['roofline']['achieved'])
for s in d.get('secondary',[]): print(s)
PY
