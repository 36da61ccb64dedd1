set -e
mkdir -p gpurun_out/tr2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CCGP_OVERLAP=1 CCGP_BENCH_NOTIMING=1 timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/tr2/o1 -o t --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/tr2/o1.log 2>&1
