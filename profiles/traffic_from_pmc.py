"""HBM bytes per conv launch from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
`python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-small`, with the calibration factors MEASURED in the conv
engine's own access patterns (profiles/traffic_calib.py, tools/traffic_calib.hip; MI355X_MICROARCH.md's factor 2.0 for
FETCH_SIZE is for 16 B/lane streaming reads), next to the ALGORITHMIC bytes of every launch
    4 * (Cin * L + Cout * L * (1 + has_residual + accumulate))         (fused ResBlock pair: 4 * C * L * (2 + accumulate);
    bf16 octet engine: the same with 2 bytes per element)
taken from the library's own launch log (TTSAMD_CONV_LOG, launch order = dispatch order).
usage: python profiles/traffic_from_pmc.py <fetch_dir> <write_dir> <conv_log.csv> <frames> [calib.json] > profiles/rN/traffic.json"""
import csv
import glob
import json
import os
import sys
from collections import OrderedDict


def dispatches(d, counter):
    """[(dispatch id, kernel, grid, KB)] of the conv launches, in dispatch order"""
    rows = {}
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter and any(k in r['Kernel_Name'] for k in ('conv1d_mfma', 'conv1d_wino', 'resblock_pair', 'resblock_chain', 'convt_mfma', 'bfo_conv1d', 'bfo_convt', 'bfo3_conv1d', 'bfo3_convt')):
                k = int(r['Dispatch_Id'])
                name = r['Kernel_Name'].split('(')[0].replace('void ttsamd::', '')
                g = None
                if 'Grid_Size_X' in r and r.get('Workgroup_Size_X'):
                    g = (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
                elif 'Grid_Size' in r:
                    g = int(r['Grid_Size'])
                v = rows.setdefault(k, [name, g, 0.0])
                v[2] += float(r['Counter_Value'])
    return [(k,) + tuple(v) for k, v in sorted(rows.items())]


def main():
    fetch, write = dispatches(sys.argv[1], 'FETCH_SIZE'), dispatches(sys.argv[2], 'WRITE_SIZE')
    log = [ln.strip().split(',') for ln in open(sys.argv[3]) if ln.strip()]
    frames = int(sys.argv[4])
    fr, fw = 2.0, 1.0
    calib_src = 'MI355X_MICROARCH.md (2.0 for 16 B/lane reads); uncalibrated for this access pattern'
    if len(sys.argv) > 5:
        c = json.load(open(sys.argv[5]))['true_bytes_per_counted_byte']
        fr = c['fetch'].get('calib_read4', fr)
        fw = c['write'].get('calib_write16', fw)
        calib_src = f'{sys.argv[5]}: calib_read4 (activation staging pattern) {fr:.3f}, calib_write16 (row epilogue) {fw:.3f}'
    n = min(len(fetch), len(write))
    # the log holds warm-up + timed step; the PMC passes hold the same launches: align from the END
    log = log[-n:] if len(log) >= n else log
    per = OrderedDict()
    tot_meas = tot_alg = 0.0
    for i in range(n):
        name, grid, fkb = fetch[-n + i][1:]
        wkb = write[-n + i][3]
        meas = (fr * fkb + fw * wkb) * 1024.0
        alg = None
        if i < len(log):
            kind = log[i][0]
            K, cin, cout, nout, batch, has_res, mode, len_mul, ragged, n_phase = map(int, log[i][1:])
            L = frames * len_mul if ragged else nout * batch          # valid positions summed over the batch
            if kind.startswith('fused_pair'):                          # resblock_pair / resblock_pair2 (256- and 128-column blocks)
                alg = 4.0 * cin * L * (2 + (mode != 0))
            elif kind in ('bfo_pair', 'bfo_chain'):                   # bf16 octet engine: bf16 tensors, one read + one write per pair / chained ResBlock
                alg = 2.0 * cin * L * (2 + (mode != 0))
            elif kind == 'bfo':
                alg = 2.0 * (cin * L + cout * L * (1 + has_res + (mode != 0)))
            elif kind == 'bfo_convt':
                alg = 2.0 * (cin * L + cout * L * n_phase)
            else:
                alg = 4.0 * (cin * L + cout * L * n_phase * (1 + has_res + (mode != 0)))
        key = f'{name} grid{grid}'
        e = per.setdefault(key, {'launches': 0, 'measured_bytes': 0.0, 'algorithmic_bytes': 0.0})
        e['launches'] += 1
        e['measured_bytes'] += meas
        e['algorithmic_bytes'] += alg or 0.0
        tot_meas += meas
        tot_alg += alg or 0.0
    for e in per.values():
        e['measured_bytes'] /= e['launches']
        e['algorithmic_bytes'] /= e['launches']
        e['ratio'] = e['measured_bytes'] / e['algorithmic_bytes'] if e['algorithmic_bytes'] else None
    out = {'calibration': calib_src, 'conv_launches': n,
           'bytes_per_conv_launch_corrected': tot_meas / max(n, 1), 'algorithmic_bytes_per_conv_launch': tot_alg / max(n, 1),
           'ratio': tot_meas / tot_alg if tot_alg else None, 'per_instantiation': per,
           'note': 'measured = factor_read * FETCH_SIZE + factor_write * WRITE_SIZE (KB -> bytes), per launch, separate PMC passes; '
                   'weights (L2-resident, float4) and halo re-reads are in "measured" but not in "algorithmic"'}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
