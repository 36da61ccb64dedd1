set -e
mkdir -p gpurun_out/sm
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not n4096 and not n8192 and not blocked" > gpurun_out/sm/pytest.log 2>&1 || { tail -40 gpurun_out/sm/pytest.log; exit 1; }
tail -2 gpurun_out/sm/pytest.log
python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/sm/b.json 2>gpurun_out/sm/b.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/sm/b.json').read().strip().splitlines()[-1])
print('ms/step %.2f'%d['ms_per_step'])
for s in d['secondary']: print(s['workload'][:40], round(s['ms_per_pass'],2))
PY
