set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02e
mkdir -p $OUT
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -60 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
timeout -k 10 600 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -20 $OUT/bench_default.err; exit 1; }
python3 - <<'PY'
import json,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r02e/bench_default.json").read().strip().splitlines()[-1])
print("value %.1f ms/step %.2f roof %.3f whole %.1f"%(d["value"],d["ms_per_step"],d["roofline"]["frac"],d["whole_job_tflops"]))
print(json.dumps(d["cpu_baseline"],indent=1))
for s in d["secondary"]: print(s["workload"][:40], s["value"], s.get("cpu"))
PY
