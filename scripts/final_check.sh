set -e
mkdir -p gpurun_out/final2
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/final2/pytest.log 2>&1 || { tail -40 gpurun_out/final2/pytest.log; exit 1; }
tail -3 gpurun_out/final2/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/profile_round.sh r01i > gpurun_out/final2/profile.log 2>&1 || { tail -20 gpurun_out/final2/profile.log; exit 1; }
tail -c 400 gpurun_out/r01i/bench_default.json
