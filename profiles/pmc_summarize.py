"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel name/grid:
    python profiles/pmc_summarize.py <dir or csv> COUNTER [COUNTER...]
prints per-kernel averages and the ratios MFMA_BUSY / (BUSY_CU_CYCLES) when present."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    src = sys.argv[1]
    files = [src] if src.endswith('.csv') else glob.glob(os.path.join(src, '**', '*counter_collection.csv'), recursive=True)
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            key = (r['Kernel_Name'][:70], r.get('Grid_Size', ''))
            acc[key][r['Counter_Name']] += float(r['Counter_Value'])
            n[key].add(r['Dispatch_Id'])
    names = sorted({c for v in acc.values() for c in v})
    print('# counters (sum over all SEs/CUs, averaged per dispatch): ' + ', '.join(names))
    rows = sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CU_CYCLES', kv[1].get(names[0], 0)))
    for key, v in rows[:40]:
        d = max(1, len(n[key]))
        line = f'{key[0]:70s} grid={key[1]:>9s} x{d:3d} ' + ' '.join(f'{c}={v[c] / d:.4g}' for c in names)
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in v and v.get('SQ_BUSY_CU_CYCLES'):
            line += f"  MfmaUtil={100 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['SQ_BUSY_CU_CYCLES'] * 4):.1f}%(/4 SIMD)"
        print(line)


if __name__ == '__main__':
    main()
