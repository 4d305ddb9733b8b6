#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV by (kernel, grid): calls, total/avg duration.
usage: python profiles/summarize.py <kernel_trace.csv> > profiles/<name>_by_grid.txt"""
import csv
import sys
from collections import defaultdict

rows = defaultdict(lambda: [0, 0])
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        key = (r['Kernel_Name'][:100], int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']),
               int(r['Grid_Size_Y']), int(r['Grid_Size_Z']), r['VGPR_Count'], r['Accum_VGPR_Count'], r['LDS_Block_Size'])
        d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        rows[key][0] += 1
        rows[key][1] += d
tot = sum(v[1] for v in rows.values())
print(f'total kernel time {tot / 1e6:.3f} ms')
print(f'{"total_ms":>10} {"calls":>6} {"avg_us":>10} {"%":>6}  blocks(x,y,z) vgpr agpr lds  kernel')
for k, v in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f'{v[1] / 1e6:10.3f} {v[0]:6d} {v[1] / v[0] / 1e3:10.1f} {100 * v[1] / tot:6.2f}  ({k[1]},{k[2]},{k[3]}) {k[4]} {k[5]} {k[6]}  {k[0]}')
