set -e
mkdir -p gpurun_out/spread
CCGP_SPREAD=1 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "blocked_path or n4096_against or cutover" > gpurun_out/spread/pytest.log 2>&1 || { tail -30 gpurun_out/spread/pytest.log; exit 1; }
tail -2 gpurun_out/spread/pytest.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
for cfg in "0 0 0" "1 0 0" "0 1 0" "1 1 0" "1 0 8" "1 1 8" "1 2 0" "0 2 0"; do
  set -- $cfg
  CCGP_SPREAD=$1 CCGP_STRIPS=$2 CCGP_WAVES=$3 timeout -k 10 200 $B > gpurun_out/spread/sp$1_s$2_w$3.json 2>gpurun_out/spread/sp$1_s$2_w$3.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/spread/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, 'ms/step %.2f'%d['ms_per_step'], {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()}, 'TF %.1f'%d['roofline']['achieved'], d['config']['failed_evals'])
    except Exception as e: print(f,'ERR',e)
PY
