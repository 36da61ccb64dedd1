# Rehearsal of the N > 1 path on a ONE-GPU box, invoked the way the driver invokes N = 1: plain `python3 bench.py
# --gpus N` (no torchrun on the command line) -- bench.py starts its own ranks.  --backend gloo: the ranks share the
# one device and the all-gather goes through host memory; RCCL needs one device per rank.
set -e
mkdir -p gpurun_out/n2
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {   # name, then bench.py arguments
  local name=$1; shift
  timeout -k 10 400 python3 bench.py "$@" > gpurun_out/n2/$name.json 2> gpurun_out/n2/$name.err || { tail -20 gpurun_out/n2/$name.err; exit 1; }
  tail -1 gpurun_out/n2/$name.json | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('$name', r['n_gpus'], 'ranks', round(r['value'],1), r['unit'], round(r['ms_per_step'],2), 'ms/step', r['config'].get('parallelism'))"
}
run n2_cfg4 --gpus 2 --steps 2 --warmup 1 --backend gloo --evals-total 64 --no-cpu-baseline --no-secondary
run n2_cfg2 --gpus 2 --steps 2 --warmup 1 --backend gloo --workload cfg2
run n3_cfg3 --gpus 3 --steps 2 --warmup 1 --backend gloo --workload cfg3
run n2_cfg5 --gpus 2 --steps 2 --warmup 1 --backend gloo --workload cfg5
run n3_cfg5 --gpus 3 --steps 2 --warmup 1 --backend gloo --workload cfg5
run n1_cfg5 --gpus 1 --steps 2 --warmup 1 --workload cfg5 --no-cpu-baseline
# and the N = 1 line under a launcher, as before
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 2 --warmup 1 --evals-total 64 --no-cpu-baseline --no-secondary > gpurun_out/n2/n1_torchrun.json 2> gpurun_out/n2/n1_torchrun.err || { tail -20 gpurun_out/n2/n1_torchrun.err; exit 1; }
tail -1 gpurun_out/n2/n1_torchrun.json | cut -c1-200
