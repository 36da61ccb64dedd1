set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r02i}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-secondary --warmup 1 --steps 1 --evals-total 64"
for V in "s1_f1:--strips 1" "s2_f1:--strips 2" "s1_f0:--strips 1 --no-fuse-diag" "s2_f0:--strips 2 --no-fuse-diag"; do
  N=${V%%:*}; A=${V#*:}
  timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/tr64_$N -o t --output-format csv -- $B $A > $OUT/tr64_$N.log 2>&1
done
B5="python3 $R/bench.py --no-cpu-baseline --no-secondary --warmup 1 --steps 1"
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/tr512_f1 -o t --output-format csv -- $B5 > $OUT/tr512_f1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/tr512_f0 -o t --output-format csv -- $B5 --no-fuse-diag > $OUT/tr512_f0.log 2>&1
ls $OUT
