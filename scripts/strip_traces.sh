# per-launch durations of the update kernels with the strip count pinned (CCGP_STRIPS = 1 | 2 | 4) and with
# the product selection: input for the cost model in pick_strips (blocked.hip)
set -e
mkdir -p gpurun_out/strips
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for S in 0 1 2 4; do
  CCGP_STRIPS=$S CCGP_BENCH_NOTIMING=1 timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/strips/s$S -o t --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/strips/s$S.log 2>&1
done
