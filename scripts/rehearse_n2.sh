set -e
mkdir -p gpurun_out/n2
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --evals-total 64 --no-cpu-baseline --no-secondary > gpurun_out/n2/n2_gloo.json 2> gpurun_out/n2/n2_gloo.err || { tail -20 gpurun_out/n2/n2_gloo.err; exit 1; }
tail -1 gpurun_out/n2/n2_gloo.json
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/n2/n1_torchrun.json 2> gpurun_out/n2/n1_torchrun.err || { tail -20 gpurun_out/n2/n1_torchrun.err; exit 1; }
tail -1 gpurun_out/n2/n1_torchrun.json
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 3 --steps 2 --warmup 1 --backend gloo --workload cfg2 > gpurun_out/n2/n3_cfg2.json 2> gpurun_out/n2/n3_cfg2.err || { tail -20 gpurun_out/n2/n3_cfg2.err; exit 1; }
tail -1 gpurun_out/n2/n3_cfg2.json
