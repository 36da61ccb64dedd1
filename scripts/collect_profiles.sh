# Copy what is to be judged from gpurun_out/<tag>/ (scratch) into profiles/<tag>/ (tracked) and run the PMC summaries.
# usage: bash scripts/collect_profiles.sh r04
set -e
TAG=${1:-r04}
G=gpurun_out/$TAG
P=profiles/$TAG
mkdir -p $P
for f in bench_default bench_slice64 bench_cfg2 bench_cfg3 bench_cfg3_grid16 bench_cfg5; do [ -f $G/$f.json ] && tail -1 $G/$f.json > $P/$f.json; done
cp $G/stats/t_kernel_stats.csv $P/kernel_stats_stats.csv
cp $G/stats64/t_kernel_stats.csv $P/kernel_stats_stats64.csv
cp $G/statsfc/t_kernel_stats.csv $P/kernel_stats_stats_fused_cov.csv
cp $G/fused_cov_ab.txt $P/fused_cov_ab.txt
cp $G/stats_cfg2/t_kernel_stats.csv $P/kernel_stats_stats_cfg2.csv
cp $G/stats_cfg3/t_kernel_stats.csv $P/kernel_stats_stats_cfg3.csv
cp $G/valu_f64_rates.txt $P/valu_f64_rates.txt
cp $G/mfma_valu_overlap.txt $P/mfma_valu_overlap.txt
python3 scripts/pmc_summary.py $TAG > /dev/null
python3 scripts/pmc_summary.py $TAG 64 64 > /dev/null
python3 scripts/pmc_summary.py $TAG fc 512 > /dev/null
python3 scripts/pmc_sq_summary.py $TAG > /dev/null
python3 scripts/pmc_cfg2_summary.py $TAG cfg2 > /dev/null
python3 scripts/pmc_cfg2_summary.py $TAG cfg3 > /dev/null
python3 scripts/pmc_cov_summary.py $TAG > /dev/null
python3 - <<PY
import csv
rows = list(csv.DictReader(open('$G/pmc_valu_clock/t_counter_collection.csv')))
with open('$P/pmc_valu_clock.txt', 'w') as out:
    out.write("clock under the fp64 VALU probe from the counters: rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -- tests/hip/valu_rates pmc\n")
    out.write("(GRBM_GUI_ACTIVE is summed over the 8 XCDs; clock = counter / 8 / kernel duration; five launches per kernel, 4 waves per SIMD)\n")
    for r in rows:
        dur = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        out.write("%-12s %8.3f ms  %.3f GHz\n" % (r['Kernel_Name'].split('(')[0], dur / 1e6, float(r['Counter_Value']) / 8 / dur))
PY
ls $P
