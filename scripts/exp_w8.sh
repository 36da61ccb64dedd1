set -e
mkdir -p gpurun_out/w8
export CCGP_WAVES=8
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "blocked_path or n4096_against or cutover" > gpurun_out/w8/pytest_w8.log 2>&1 || { tail -30 gpurun_out/w8/pytest_w8.log; exit 1; }
tail -3 gpurun_out/w8/pytest_w8.log
unset CCGP_WAVES
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
timeout -k 10 200 $B > gpurun_out/w8/base.json 2>gpurun_out/w8/base.err
CCGP_WAVES=8 timeout -k 10 200 $B > gpurun_out/w8/w8_auto.json 2>gpurun_out/w8/w8_auto.err
CCGP_WAVES=8 CCGP_STRIPS=1 timeout -k 10 200 $B > gpurun_out/w8/w8_s1.json 2>gpurun_out/w8/w8_s1.err
CCGP_WAVES=8 CCGP_STRIPS=2 timeout -k 10 200 $B > gpurun_out/w8/w8_s2.json 2>gpurun_out/w8/w8_s2.err
CCGP_STRIPS=1 timeout -k 10 200 $B > gpurun_out/w8/w4_s1.json 2>gpurun_out/w8/w4_s1.err
CCGP_STRIPS=2 timeout -k 10 200 $B > gpurun_out/w8/w4_s2.json 2>gpurun_out/w8/w4_s2.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/w8/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, 'ms/step %.2f'%d['ms_per_step'], {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()}, 'TF %.1f'%d['roofline']['achieved'])
    except Exception as e: print(f,'ERR',e)
PY
