set -e
mkdir -p gpurun_out/ov2
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary"
for cfg in "0 1 0" "1 1 0" "1 1 1" "1 2 1" "1 4 1"; do
  set -- $cfg
  CCGP_OVERLAP=$1 CCGP_DIAG_S=$2 CCGP_AUX_PRIO=$3 timeout -k 10 200 $B > gpurun_out/ov2/o$1_s$2_p$3.json 2>gpurun_out/ov2/o$1_s$2_p$3.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ov2/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, 'ms/step %.2f'%d['ms_per_step'], {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()}, d['config']['failed_evals'], d['config']['all_finite'])
    except Exception as e: print(f,'ERR',e)
PY
