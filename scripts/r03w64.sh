# per-launch table of the 64-matrix slice (one GPU's share at N = 8): kernel trace of one step
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r03w64}
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python3 bench.py --evals-total 64 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
cp $(find $OUT/trace -name 't_kernel_trace.csv' | head -1) $OUT/kernel_trace.csv
rm -rf $OUT/trace
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
covs=[i for i,r in enumerate(rows) if 'cov_kernel' in r['Kernel_Name']]
seg=rows[covs[-1]:]
j=0; jt=0; tu=0; tt=0
for r in seg:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    wg=int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])
    if 'chol_update' in r['Kernel_Name']:
        j+=1; tu+=d
        tiles=64*(31-j); w1=64+tiles
        print('update j=%2d  %7.1f us  %5d wg  (S=1 count %4d = %.2f rounds of 256)  us per round-of-K128: %.2f' % (j,d,wg,w1,w1/256.0,d/j/max(1,-(-w1//256))))
    elif 'chol_trsm' in r['Kernel_Name']:
        tt+=d; print('   trsm j=%2d  %6.1f us %5d wg' % (jt,d,wg)); jt+=1
    elif 'ccgp' in r['Kernel_Name']:
        print('   ', r['Kernel_Name'][:50].split('(')[0][-30:], round(d,1))
print('update total', tu/1e3, 'ms; trsm total', tt/1e3)
PY
