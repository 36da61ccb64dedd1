set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r02o}
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_factorset.py tests/test_gpu_multi.py -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
B="python3 $R/bench.py --no-cpu-baseline --no-secondary --warmup 1"
timeout -k 10 200 $B --steps 10 --evals-total 64 > $OUT/b64.json 2> $OUT/b64.err
timeout -k 10 200 $B --steps 4 > $OUT/b512.json 2> $OUT/b512.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/tr64 -o t --output-format csv -- $B --steps 1 --evals-total 64 > $OUT/tr64.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/tr512 -o t --output-format csv -- $B --steps 1 > $OUT/tr512.log 2>&1
cd $R
python3 - $OUT <<'PY'
import json,glob,os,sys,csv
for f in sorted(glob.glob(sys.argv[1]+"/b*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), "value %.1f ms/step %.2f roof %.3f"%(d["value"],d["ms_per_step"],d["roofline"]["frac"]), {k:round(v,2) for k,v in d["kernel_ms_per_step"].items()}, d["config"].get("matches_cpu_potrf_digest"))
def tab(p):
    rows=list(csv.DictReader(open(p)))
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    covs=[i for i,r in enumerate(rows) if 'cov_kernel' in r['Kernel_Name']]
    seg=rows[covs[-1]:]
    return [(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in seg if 'chol_update' in r['Kernel_Name']]
for n in ("tr64","tr512"):
    t=tab(sys.argv[1]+"/%s/t_kernel_trace.csv"%n)
    print(n, "sum %.2f ms"%(sum(t)/1e3), "late:", [round(x) for x in t[23:]])
PY
