# Copy what is to be judged from gpurun_out/<tag>/ (scratch) into profiles/<tag>/ (tracked) and run the PMC summaries.
# usage: bash scripts/collect_profiles_r05.sh r05
set -e
TAG=${1:-r05}
G=gpurun_out/$TAG
P=profiles/$TAG
mkdir -p $P
for f in bench_default bench_slice64 bench_slice64_sched0 bench_slice64_b bench_slice64_sched0_b bench_sched1 bench_cfg2 bench_cfg3 bench_cfg5; do [ -s $G/$f.json ] && tail -1 $G/$f.json > $P/$f.json; done
for f in $G/sched_account_*.json; do [ -s $f ] && cp $f $P/; done
cp $G/stats/t_kernel_stats.csv $P/kernel_stats_stats.csv
cp $G/statss1/t_kernel_stats.csv $P/kernel_stats_stats_sched1.csv
cp $G/stats64/t_kernel_stats.csv $P/kernel_stats_stats64.csv
cp $G/stats64s0/t_kernel_stats.csv $P/kernel_stats_stats64_sched0.csv
cp $G/stats_cfg5/t_kernel_stats.csv $P/kernel_stats_stats_cfg5.csv
cp $G/stats_cfg2/t_kernel_stats.csv $P/kernel_stats_stats_cfg2.csv
cp $G/stats_cfg3/t_kernel_stats.csv $P/kernel_stats_stats_cfg3.csv
python3 scripts/pmc_summary.py $TAG > /dev/null
python3 scripts/pmc_summary.py $TAG s1 512 > /dev/null
python3 scripts/pmc_summary.py $TAG 64 64 > /dev/null
python3 scripts/pmc_summary.py $TAG 64s0 64 > /dev/null
python3 scripts/pmc_sq_summary.py $TAG > /dev/null
ls $P
