set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r02m}
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_factorset.py tests/test_gpu_multi.py -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
B="python3 $R/bench.py --no-cpu-baseline --no-secondary --warmup 1"
for T in "" "--no-tail-strips"; do
  N=t1; [ -n "$T" ] && N=t0
  timeout -k 10 200 $B --steps 10 --evals-total 64 $T > $OUT/b64_$N.json 2> $OUT/b64_$N.err
  timeout -k 10 200 $B --steps 6 --evals-total 128 $T > $OUT/b128_$N.json 2> $OUT/b128_$N.err
  timeout -k 10 200 $B --steps 4 --evals-total 256 $T > $OUT/b256_$N.json 2> $OUT/b256_$N.err
  timeout -k 10 200 $B --steps 4 $T > $OUT/b512_$N.json 2> $OUT/b512_$N.err
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/tr64_t1 -o t --output-format csv -- $B --steps 1 --evals-total 64 > $OUT/tr64_t1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/tr64_t0 -o t --output-format csv -- $B --steps 1 --evals-total 64 --no-tail-strips > $OUT/tr64_t0.log 2>&1
cd $R
python3 - $OUT <<'PY'
import json,glob,os,sys
for f in sorted(glob.glob(sys.argv[1]+"/b*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), "value %.1f ms/step %.2f roof %.3f"%(d["value"],d["ms_per_step"],d["roofline"]["frac"]), {k:round(v,2) for k,v in d["kernel_ms_per_step"].items()}, d["config"].get("matches_cpu_potrf_digest"))
PY
