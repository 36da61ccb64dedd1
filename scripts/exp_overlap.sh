set -e
mkdir -p gpurun_out/ovl
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "blocked_path or n4096 or cutover or non_positive" > gpurun_out/ovl/pytest.log 2>&1 || { tail -30 gpurun_out/ovl/pytest.log; exit 1; }
tail -2 gpurun_out/ovl/pytest.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
for O in 0 1; do
  CCGP_OVERLAP=$O timeout -k 10 200 $B > gpurun_out/ovl/o$O.json 2>gpurun_out/ovl/o$O.err
  CCGP_OVERLAP=$O CCGP_BENCH_NOTIMING=1 timeout -k 10 200 $B > gpurun_out/ovl/o${O}_nt.json 2>gpurun_out/ovl/o${O}_nt.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ovl/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, 'ms/step %.2f'%d['ms_per_step'], {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()}, 'TF %.1f'%d['roofline']['achieved'], d['config']['failed_evals'])
    except Exception as e: print(f,'ERR',e)
PY
