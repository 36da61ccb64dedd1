# same box: scheduler with two workgroups per CU and different backlog thresholds for the second one (CCGP_SCHED_BACKLOG), against
# one workgroup per CU (--sched 2) and the launches (--sched 0)
mkdir -p gpurun_out/r05n
one() { # label, env, args
  lab=$1; shift; envs=$1; shift
  env $envs python bench.py --no-cpu-baseline --no-secondary --steps 8 --warmup 2 "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lab', round(d['ms_per_step'],2), d['config']['matches_cpu_potrf_digest'])"
}
for B in 64 128 256; do
  one "B=$B launches" "X=1" --evals-total $B --sched 0
  one "B=$B sched2" "X=1" --evals-total $B --sched 2
  for T in 32 64 128 256 512; do one "B=$B sched1 backlog=$T" "CCGP_SCHED_BACKLOG=$T" --evals-total $B --sched 1; done
done
