# round 2c: fused diagonal + right-hand-side tile
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r02c}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
B="python3 $R/bench.py --no-cpu-baseline --no-secondary --warmup 1"
for S in 0 1 2 4; do
  timeout -k 10 200 $B --steps 4 --strips $S > $OUT/b512_s${S}.json 2> $OUT/b512_s${S}.err
  timeout -k 10 200 $B --steps 10 --strips $S --evals-total 64 > $OUT/b64_s${S}.json 2> $OUT/b64_s${S}.err
done
timeout -k 10 200 $B --steps 6 --evals-total 128 > $OUT/b128_s0.json 2> $OUT/b128_s0.err
timeout -k 10 200 $B --steps 6 --evals-total 256 > $OUT/b256_s0.json 2> $OUT/b256_s0.err
cd /tmp && export TMPDIR=/tmp
for S in 1 2 4; do
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/tr512_s$S -o t --output-format csv -- $B --steps 1 --strips $S > $OUT/tr512_s$S.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/tr64_s$S -o t --output-format csv -- $B --steps 1 --strips $S --evals-total 64 > $OUT/tr64_s$S.log 2>&1
done
cd $R
python3 - $TAG <<'PY'
import json,glob,os,sys
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/%s/b*.json"%sys.argv[1])):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "value %.1f ms/step %.2f roof %.3f whole %.1f"%(d["value"],d["ms_per_step"],d["roofline"]["frac"],d["whole_job_tflops"]), {k:round(v,2) for k,v in d["kernel_ms_per_step"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
