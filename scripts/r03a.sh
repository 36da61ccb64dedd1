# Round-3 first GPU pass: the whole GPU suite, the N > 1 rehearsal (bench.py starts its own ranks), the default bench line.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03a
mkdir -p $OUT
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -3 $OUT/pytest_gpu.log
bash scripts/rehearse_n2.sh
timeout -k 10 600 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -20 $OUT/bench_default.err; exit 1; }
tail -c 3000 $OUT/bench_default.json
