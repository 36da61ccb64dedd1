set -e
TAG=${1:-r01j}
mkdir -p gpurun_out/final_$TAG
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/final_$TAG/pytest.log 2>&1 || { tail -40 gpurun_out/final_$TAG/pytest.log; exit 1; }
tail -3 gpurun_out/final_$TAG/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/profile_round.sh $TAG > gpurun_out/final_$TAG/profile.log 2>&1 || { tail -20 gpurun_out/final_$TAG/profile.log; exit 1; }
bash scripts/profile_cfg2.sh $TAG > gpurun_out/final_$TAG/profile_cfg2.log 2>&1 || { tail -20 gpurun_out/final_$TAG/profile_cfg2.log; exit 1; }
python - <<PY
import json
d=json.loads(open('gpurun_out/$TAG/bench_default.json').read().strip().splitlines()[-1])
print('cfg4: %.1f evals/s %.2f ms frac %.3f'%(d['value'], d['ms_per_step'], d['roofline']['frac']), [round(s['ms_per_pass'],2) for s in d['secondary']])
d=json.loads(open('gpurun_out/${TAG}_cfg2/bench_cfg2.json').read().strip().splitlines()[-1])
print('cfg2: %.3g evals/s %.2f ms'%(d['value'], d['ms_per_step']), d['cpu_baseline'])
PY
