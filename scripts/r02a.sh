# round 2a: first measurement of the whole 512-point grid as one chunk on one GPU
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02a
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
B="python3 $R/bench.py --no-cpu-baseline --no-secondary"
timeout -k 10 300 $B --steps 5 --warmup 2 > $OUT/b512.json 2> $OUT/b512.err
timeout -k 10 300 $B --steps 5 --warmup 2 --strips 1 > $OUT/b512_s1.json 2> $OUT/b512_s1.err
timeout -k 10 300 $B --steps 5 --warmup 2 --evals-total 256 > $OUT/b256.json 2> $OUT/b256.err
timeout -k 10 300 $B --steps 5 --warmup 2 --evals-total 128 > $OUT/b128.json 2> $OUT/b128.err
timeout -k 10 300 $B --steps 10 --warmup 2 --evals-total 64 > $OUT/b64.json 2> $OUT/b64.err
timeout -k 10 300 $B --steps 5 --warmup 2 --ws-limit-gib 20 > $OUT/b512_ws20.json 2> $OUT/b512_ws20.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o t --output-format csv -- $B --steps 2 --warmup 1 > $OUT/stats.log 2>&1
cd $R
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r02a/b*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "value %.1f ms/step %.2f roof %.3f whole %.1f"%(d["value"],d["ms_per_step"],d["roofline"]["frac"],d["whole_job_tflops"]), d["kernel_ms_per_step"])
    except Exception as e:
        print(f, "ERR", e)
PY
