"""Summarise a rocprofv3 --kernel-trace CSV by (kernel, grid): calls, total/avg duration.
usage: python profiles/summarize.py <kernel_trace.csv> > profiles/<name>_by_grid.txt"""
import csv
import sys
from collections import defaultdict

rows = defaultdict(lambda: [0, 0])
conv_iv = []          # (start, end) of every MFMA conv launch: their UNION is the conv engine's busy time
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if any(k in r['Kernel_Name'] for k in ('conv1d_mfma', 'conv1d_wino', 'resblock_pair', 'resblock_chain', 'convt_mfma', 'bfo_conv1d', 'bfo_convt', 'bfo3_conv1d', 'bfo3_convt')):
            conv_iv.append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
        key = (r['Kernel_Name'][:100], int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']),
               int(r['Grid_Size_Y']), int(r['Grid_Size_Z']), r['VGPR_Count'], r['Accum_VGPR_Count'], r['LDS_Block_Size'])
        d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        rows[key][0] += 1
        rows[key][1] += d
tot = sum(v[1] for v in rows.values())
print(f'total kernel time {tot / 1e6:.3f} ms')
if conv_iv:
    conv_iv.sort()
    union, cur_s, cur_e = 0, conv_iv[0][0], conv_iv[0][1]
    for a, b in conv_iv[1:]:
        if a > cur_e:
            union += cur_e - cur_s
            cur_s, cur_e = a, b
        else:
            cur_e = max(cur_e, b)
    union += cur_e - cur_s
    print(f'MFMA conv launches (conv1d_mfma / conv1d_wino / resblock_pair / convt_mfma / bfo_* / bfo3_*): {len(conv_iv)}, sum of durations {sum(b - a for a, b in conv_iv) / 1e6:.3f} ms, '
          f'union of their intervals (busy time of the conv engine; the three ResBlock branches of a HiFi-GAN stage '
          f'overlap on three streams) {union / 1e6:.3f} ms')
print(f'{"total_ms":>10} {"calls":>6} {"avg_us":>10} {"%":>6}  blocks(x,y,z) vgpr agpr lds  kernel')
for k, v in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f'{v[1] / 1e6:10.3f} {v[0]:6d} {v[1] / v[0] / 1e3:10.1f} {100 * v[1] / tot:6.2f}  ({k[1]},{k[2]},{k[3]}) {k[4]} {k[5]} {k[6]}  {k[0]}')
