# round 2b: inter-matrix skew (HBM channel aliasing) experiment
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02b
mkdir -p $OUT
cd $R
B="python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 4 --warmup 1"
for SK in 0 544 2080 8224; do
  for S in 1 0; do
    timeout -k 10 200 $B --strips $S --skew $SK > $OUT/b512_s${S}_k${SK}.json 2> $OUT/b512_s${S}_k${SK}.err
  done
  timeout -k 10 200 $B --skew $SK --evals-total 64 --steps 10 > $OUT/b64_s0_k${SK}.json 2> $OUT/b64_s0_k${SK}.err
done
cd /tmp && export TMPDIR=/tmp
for SK in 0 544; do
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/tr_k$SK -o t --output-format csv -- $B --steps 1 --strips 1 --skew $SK > $OUT/tr_k$SK.log 2>&1
done
cd $R
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r02b/b*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "value %.1f ms/step %.2f roof %.3f whole %.1f"%(d["value"],d["ms_per_step"],d["roofline"]["frac"],d["whole_job_tflops"]), {k:round(v,2) for k,v in d["kernel_ms_per_step"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
