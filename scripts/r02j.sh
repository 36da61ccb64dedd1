set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r02j}
mkdir -p $OUT
cd $R
B="python3 $R/bench.py --no-cpu-baseline --no-secondary --warmup 1 --strips 1"
for D in 0 1; do
  timeout -k 10 200 $B --steps 4 --deep $D > $OUT/b512_d$D.json 2> $OUT/b512_d$D.err
  timeout -k 10 200 $B --steps 10 --evals-total 64 --deep $D > $OUT/b64_d$D.json 2> $OUT/b64_d$D.err
  timeout -k 10 200 $B --steps 6 --evals-total 128 --deep $D > $OUT/b128_d$D.json 2> $OUT/b128_d$D.err
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/tr64_d1 -o t --output-format csv -- $B --steps 1 --evals-total 64 --deep 1 > $OUT/tr64_d1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/tr512_d1 -o t --output-format csv -- $B --steps 1 --deep 1 > $OUT/tr512_d1.log 2>&1
cd $R
python3 - $OUT <<'PY'
import json,glob,os,sys
for f in sorted(glob.glob(sys.argv[1]+"/b*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), "value %.1f ms/step %.2f roof %.3f"%(d["value"],d["ms_per_step"],d["roofline"]["frac"]), {k:round(v,2) for k,v in d["kernel_ms_per_step"].items()}, d["config"].get("matches_cpu_potrf_digest"))
PY
