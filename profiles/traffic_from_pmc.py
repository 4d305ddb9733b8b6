"""HBM bytes per conv launch from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
`python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline`, corrected as MI355X_MICROARCH.md prescribes for gfx950
(both counters are in KB; FETCH_SIZE is doubled).  usage:
    python profiles/traffic_from_pmc.py <fetch_dir> <write_dir> > profiles/rN/traffic.json"""
import csv
import glob
import json
import os
import sys


def conv_sum(d, counter):
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    tot, disp = 0.0, set()
    for f in files:
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter and 'conv1d_mfma' in r['Kernel_Name']:
                tot += float(r['Counter_Value'])
                disp.add(r['Dispatch_Id'])
    return tot, len(disp)


def main():
    fetch, nf = conv_sum(sys.argv[1], 'FETCH_SIZE')
    write, nw = conv_sum(sys.argv[2], 'WRITE_SIZE')
    n = max(nf, nw, 1)
    out = {'FETCH_SIZE': {'sum_kb_over_conv_launches': fetch, 'launches': nf},
           'WRITE_SIZE': {'sum_kb_over_conv_launches': write, 'launches': nw},
           'bytes_per_conv_launch_corrected': (2.0 * fetch + write) * 1024.0 / n,
           'note': 'FETCH_SIZE doubled (gfx950 rocprofv3 reports half of a coalesced streaming read, MI355X_MICROARCH.md '
                   'HBM section; calibrated for 16 B/lane loads, our activation loads are 4 B/lane so the read side is an '
                   'upper estimate); WRITE_SIZE matches the algorithmic write bytes (459 MB for a C=128 stage conv).'}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
