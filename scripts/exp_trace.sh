set -e
mkdir -p gpurun_out/tr
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "12 1" "12 0" "0 1"; do
  set -- $cfg
  CCGP_ABLATE=$1 CCGP_STRIPS=$2 CCGP_BENCH_NOTIMING=1 timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/tr/a$1_s$2 -o t --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/tr/a$1_s$2.log 2>&1
done
ls -R $R/gpurun_out/tr | head -30
