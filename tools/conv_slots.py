"""Ragged-batch timeline analysis of a -DTTS_TIMING conv_bench CSV: live vs dead blocks (durations, counts), and the
dispatch order effect: for every block the delay between the previous block's start (linear block id order) and its own.
    python tools/conv_slots.py gpurun_out/timing_c128_k7.csv"""
import csv
import sys

import numpy as np

TICK_US = 1e-2
for fn in sys.argv[1:]:
    rows = list(csv.DictReader(open(fn)))
    blk = np.array([int(r['block']) for r in rows])
    st = np.array([int(r['start']) for r in rows], dtype=np.float64) * TICK_US
    en = np.array([int(r['end']) for r in rows], dtype=np.float64) * TICK_US
    mn = np.array([int(r['main']) for r in rows], dtype=np.float64)
    dead = mn == 0
    t0 = st.min()
    print(f'{fn}: {len(rows)} stamped blocks, {dead.sum()} dead; span {en.max() - t0:.0f} us')
    print(f'  live: mean duration {np.mean((en - st)[~dead]):.1f} us;  dead: mean {np.mean((en - st)[dead]) if dead.any() else 0:.2f} us, '
          f'max {np.max((en - st)[dead]) if dead.any() else 0:.2f} us')
    order = np.argsort(blk)
    s_sorted = st[order] - t0
    d_sorted = dead[order]
    # start-time profile along the linear block id: how long does the dispatcher sit on runs of dead blocks?
    gaps = np.diff(s_sorted)
    print(f'  start-to-start gap along block ids: live->live median {np.median(gaps[~d_sorted[1:] & ~d_sorted[:-1]]):.3f} us; '
          f'within dead runs median {np.median(gaps[d_sorted[1:] & d_sorted[:-1]]) if (d_sorted[1:] & d_sorted[:-1]).any() else 0:.3f} us, '
          f'mean {np.mean(gaps[d_sorted[1:] & d_sorted[:-1]]) if (d_sorted[1:] & d_sorted[:-1]).any() else 0:.3f} us')
    # slot occupancy over time: number of live blocks resident, sampled
    ts = np.linspace(0, en.max() - t0, 200)
    occ = [(np.sum((st[~dead] - t0 <= t) & (en[~dead] - t0 > t))) for t in ts]
    print('  resident live blocks over time (20 samples):', ' '.join(str(int(v)) for v in occ[::10]))

# ---- per-CU view: which CUs sit with a free slot, and for how long between two blocks -----------------------
for fn in sys.argv[1:]:
    rows = [r for r in csv.DictReader(open(fn)) if int(r['main']) != 0]
    hw = np.array([int(r['hwid']) for r in rows])
    xcc = np.array([int(r['xcc']) & 0xf for r in rows])
    cu = (hw >> 8) & 0xf
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    st = np.array([int(r['start']) for r in rows], dtype=np.float64) * TICK_US
    en = np.array([int(r['end']) for r in rows], dtype=np.float64) * TICK_US
    t0, t1 = st.min(), en.max()
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    keys = np.unique(key)
    busy = []
    for k in keys:
        m = key == k
        busy.append(np.sum(en[m] - st[m]) / (t1 - t0))
    busy = np.array(busy)
    print(f'{fn}: {len(keys)} CUs seen; resident blocks per CU: mean {busy.mean():.2f}, min {busy.min():.2f}, max {busy.max():.2f}')
    per_x = [busy[(keys // (8 * 2 * 16)) == x].mean() for x in range(8)]
    print('  per XCD:', ' '.join(f'{v:.2f}' for v in per_x))
    nblk = [int(np.sum((key // (8 * 2 * 16)) == x)) for x in range(8)]
    print('  live blocks per XCD:', nblk)
    last_end = [en[(key // (8 * 2 * 16)) == x].max() - t0 for x in range(8)]
    print('  last block end per XCD (us):', ' '.join(f'{v:.0f}' for v in last_end))
