"""GPU idle time between kernels from a rocprofv3 --kernel-trace CSV.
usage: python profiles/gaps.py <kernel_trace.csv> [min_gap_us]
Prints busy/idle totals and the gaps above min_gap_us with the kernels on either side."""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60],
                     int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']), int(r['Grid_Size_Z'])))
rows.sort()
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
busy = sum(e - s for s, e, *_ in rows)
span = rows[-1][1] - rows[0][0]
print(f'{len(rows)} kernels, span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {(span - busy) / 1e6:.3f} ms')
hist = {}
small = 0
for a, b in zip(rows, rows[1:]):
    g = (b[0] - a[1]) / 1e3
    if g < thr:
        small += max(g, 0.0)
        continue
    key = (a[2], b[2])
    h = hist.setdefault(key, [0, 0.0])
    h[0] += 1
    h[1] += g
print(f'gaps below {thr} us: {small / 1e3:.3f} ms total')
for k, v in sorted(hist.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f'{v[1] / 1e3:8.3f} ms {v[0]:4d}x avg {v[1] / v[0]:8.1f} us   {k[0]}  ->  {k[1]}')
