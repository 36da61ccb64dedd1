set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r02h}
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_factorset.py tests/test_gpu_multi.py -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
B="python3 $R/bench.py --no-cpu-baseline --no-secondary --warmup 1"
timeout -k 10 200 $B --steps 4 > $OUT/b512.json 2> $OUT/b512.err
timeout -k 10 200 $B --steps 6 --evals-total 256 > $OUT/b256.json 2> $OUT/b256.err
timeout -k 10 200 $B --steps 6 --evals-total 128 > $OUT/b128.json 2> $OUT/b128.err
timeout -k 10 200 $B --steps 6 --evals-total 128 --strips 1 > $OUT/b128_s1.json 2> $OUT/b128_s1.err
timeout -k 10 200 $B --steps 10 --evals-total 64 > $OUT/b64.json 2> $OUT/b64.err
timeout -k 10 200 $B --steps 10 --evals-total 64 --strips 1 > $OUT/b64_s1.json 2> $OUT/b64_s1.err
python3 - $OUT <<'PY'
import json,glob,os,sys
for f in sorted(glob.glob(sys.argv[1]+"/b*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), "value %.1f ms/step %.2f roof %.3f"%(d["value"],d["ms_per_step"],d["roofline"]["frac"]), {k:round(v,2) for k,v in d["kernel_ms_per_step"].items()}, d["config"].get("matches_cpu_potrf_digest"))
PY
