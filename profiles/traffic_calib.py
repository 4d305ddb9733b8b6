"""FETCH_SIZE / WRITE_SIZE calibration factors from two rocprofv3 --pmc passes of tools/bin/traffic_calib (every kernel
moves exactly 1 GiB per launch): bytes actually moved per counted KB, per access pattern.
usage: python profiles/traffic_calib.py <fetch_dir> <write_dir> > profiles/rN/traffic_calib.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

N_BYTES = 1 << 30


def per_kernel(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        by_disp = defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                by_disp[r['Dispatch_Id']] += float(r['Counter_Value'])
                names[r['Dispatch_Id']] = r['Kernel_Name'].split('(')[0]
        for k, v in by_disp.items():
            acc[names[k]].append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch, write = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
    out = {'bytes_per_launch': N_BYTES, 'FETCH_SIZE_kb': fetch, 'WRITE_SIZE_kb': write}
    f = {k: N_BYTES / (v * 1024.0) for k, v in fetch.items() if 'read' in k and v > 0}
    w = {k: N_BYTES / (v * 1024.0) for k, v in write.items() if 'write' in k and v > 0}
    out['true_bytes_per_counted_byte'] = {'fetch': f, 'write': w}
    out['note'] = ('factor = bytes the kernel really moved / (counter x 1024).  MI355X_MICROARCH.md gives 2.0 for FETCH_SIZE on 16 B/lane '
                   'streaming reads; calib_read4 is the conv engine\'s activation staging pattern (one dword per lane, 4 rows two '
                   'channels apart), calib_write4s the polyphase upsampler store pattern.')
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
