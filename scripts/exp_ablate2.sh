set -e
mkdir -p gpurun_out/abl2
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
for S in 1; do
for A in 0 16 32 48 8 17 36; do
  CCGP_ABLATE=$A CCGP_STRIPS=$S timeout -k 10 200 $B > gpurun_out/abl2/s${S}_a${A}.json 2>gpurun_out/abl2/s${S}_a${A}.err
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/abl2/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, 'ms/step %.2f'%d['ms_per_step'], {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()}, 'TF %.1f'%d['roofline']['achieved'])
    except Exception as e: print(f,'ERR',e)
PY
