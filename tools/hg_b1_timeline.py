"""Timeline of the LAST forward in a rocprofv3 kernel trace of tools/hg_b1_trace.py: start offset, duration, queue, grid, kernel."""
import csv
import glob
import os
import re
import sys

f = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?'), r['Kernel_Name'],
                 (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))))
rows.sort()
# split into forwards at gaps > 200 us
fw, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if b[0] - max(x[1] for x in cur) > 200000:
        fw.append(cur)
        cur = []
    cur.append(b)
fw.append(cur)
last = fw[-1]
t0 = last[0][0]
print(f'{len(fw)} forwards; last one: {len(last)} kernels, span {(max(x[1] for x in last) - t0) / 1e3:.1f} us, sum of durations {sum(x[1] - x[0] for x in last) / 1e3:.1f} us')
for s, e, q, n, g in last:
    short = re.sub(r'\(.*', '', n.replace('void ', '').replace('ttsamd::', ''))[:44]
    print(f'{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:6.1f} us  q{q:>3s} {str(g):16s} {short}')
