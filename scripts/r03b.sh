# exp_cov table version: GPU suite, default bench, cfg2 bench, VALU rates
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r03b}
mkdir -p $OUT
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -3 $OUT/pytest_gpu.log
timeout -k 10 600 python3 bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -20 $OUT/bench_default.err; exit 1; }
python3 - <<PY
import json
r=json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print("cfg4", round(r["value"],1), "evals/s", round(r["ms_per_step"],2), "ms", r["kernel_ms_per_step"], "K3", round(r["roofline"]["frac"],3))
for s in r["secondary"]:
    print(s["workload"][:40], round(s.get("ms_per_pass",0),3), "ms", s.get("end_to_end",{}).get("ms_per_call", s.get("end_to_end",{}).get("ms_per_pass")))
PY
timeout -k 10 300 python3 bench.py --evals-total 64 --steps 10 --no-cpu-baseline --no-secondary > $OUT/bench_slice64.json 2> $OUT/bench_slice64.err
python3 -c "
import json
r=json.loads(open('$OUT/bench_slice64.json').read().strip().splitlines()[-1])
print('slice64', round(r['ms_per_step'],2), 'ms', r['kernel_ms_per_step'], 'K3', round(r['roofline']['frac'],3))"
if [ ! -x tests/hip/valu_rates ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tests/hip/valu_f64_rates.hip -o tests/hip/valu_rates; fi
timeout -k 10 120 tests/hip/valu_rates > $OUT/valu_rates.txt; cat $OUT/valu_rates.txt
