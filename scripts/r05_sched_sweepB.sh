# same-box: launch-per-phase (sched 0) against the scheduler (1: two workgroups per CU, 2: one) over the chunk size
mkdir -p gpurun_out/r05f
for B in 8 16 32 64 96 128 192 256 384; do for S in 0 1 2; do
timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --evals-total $B --steps 6 --warmup 2 --sched $S > gpurun_out/r05f/b${B}_s$S.json 2> gpurun_out/r05f/b${B}_s$S.err || { echo FAILED $B $S; exit 1; }
python - $B $S <<'PY'
import sys, json
B, S = sys.argv[1:3]
d = json.loads(open("gpurun_out/r05f/b%s_s%s.json" % (B, S)).read().strip().splitlines()[-1])
print("B=%s sched=%s ms/step %.2f notiming %.2f" % (B, S, d["ms_per_step"], d["notiming_ms_per_step"]), {k: round(v, 2) for k, v in d["kernel_ms_per_step"].items()}, d["config"]["matches_cpu_potrf_digest"], flush=True)
PY
done; done
