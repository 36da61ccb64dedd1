#!/usr/bin/env python3
"""Per-launch table of the blocked-Cholesky update kernels from a rocprofv3 kernel trace (last cfg4 step)."""
import csv, re, sys
def table(path, nt=32, nb=64):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    covs = [i for i, r in enumerate(rows) if 'cov_kernel' in r['Kernel_Name']]
    seg = rows[covs[-1]:]
    t3 = 2.0 * 128 ** 3
    out, j = {}, 0
    for r in seg:
        if 'chol_update' in r['Kernel_Name']:
            j += 1
            dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            fl = nb * j * t3 * ((nt - 1 - j) + 0.5)
            wg = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])
            out[j] = (dur, fl / dur / 1e6, wg, re.search(r'(chol_update\w*)', r['Kernel_Name']).group(1))
        if j == nt - 1: break
    return out
if __name__ == '__main__':
    args = sys.argv[1:]
    nb = 512                       # matrices per launch (--nb 64 for the slice)
    if args and args[0] == '--nb':
        nb = int(args[1]); args = args[2:]
    tabs = [table(p, nb=nb) for p in args]
    print('j | ' + ' | '.join(args))
    for j in range(1, 32):
        print(j, ' | '.join('%7.1f us %5.1f TF %5d wg %s' % t[j] for t in tabs))
    print('total ms', [round(sum(v[0] for v in t.values()) / 1e3, 2) for t in tabs])
