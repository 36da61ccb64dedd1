# same-box A/B: launch-per-phase sweep (sched 0) against the dataflow scheduler (1 = two workgroups per CU, 2 = one), 64-matrix
# slice and full 512-point grid
mkdir -p gpurun_out/r05b
run() {  # name, args...
  name=$1; shift
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary "$@" > gpurun_out/r05b/$name.json 2> gpurun_out/r05b/$name.err || { echo "$name FAILED"; tail -5 gpurun_out/r05b/$name.err; return 1; }
  python - "$name" <<'PY'
import sys, json
name = sys.argv[1]
d = json.loads(open("gpurun_out/r05b/%s.json" % name).read().strip().splitlines()[-1])
print(name, "ms/step %.2f" % d["ms_per_step"], "notiming %.2f" % d["notiming_ms_per_step"], {k: round(v, 2) for k, v in d["kernel_ms_per_step"].items()},
      "frac %.3f" % d["roofline"]["frac"], d["config"]["matches_cpu_potrf_digest"], flush=True)
PY
}
for rep in 1 2; do
run slice_s0_$rep --evals-total 64 --steps 10 --sched 0 &&
run slice_s1_$rep --evals-total 64 --steps 10 --sched 1 &&
run slice_s1p3_$rep --evals-total 64 --steps 10 --sched 1 --sched-policy 3 &&
run slice_s2p3_$rep --evals-total 64 --steps 10 --sched 2 --sched-policy 3 &&
run slice_s2_$rep --evals-total 64 --steps 10 --sched 2 || exit 1
done
run full_s0 --steps 5 --sched 0 &&
run full_s1 --steps 5 --sched 1 &&
run full_s1p3 --steps 5 --sched 1 --sched-policy 3 &&
run full_s2 --steps 5 --sched 2 &&
run full_s0b --steps 5 --sched 0 &&
run full_s1b --steps 5 --sched 1
