set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02f
mkdir -p $OUT
cd $R
B="python3 $R/bench.py --no-cpu-baseline --no-secondary --warmup 1 --steps 3"
for V in old cg1 cg2 cg3; do
  if [ $V = cg3 ]; then unset CCGP_LIB; else export CCGP_LIB=$R/scripts/_libs/libccgp_$V.so; fi
  timeout -k 10 200 $B > $OUT/b512_$V.json 2> $OUT/b512_$V.err
  timeout -k 10 200 $B --evals-total 64 --steps 8 > $OUT/b64_$V.json 2> $OUT/b64_$V.err
done
unset CCGP_LIB
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r02f/b*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), "value %.1f ms/step %.2f"%(d["value"],d["ms_per_step"]), {k:round(v,2) for k,v in d["kernel_ms_per_step"].items()}, d["config"].get("matches_cpu_potrf_digest"))
PY
