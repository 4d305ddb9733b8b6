"""Summarise the per-block timelines dumped by `tools/bin/conv_bench` built with -DTTS_TIMING
(gpurun_out/timing_c<Cin>_k<K>.csv[.gz]): mean prologue / main-loop / epilogue time per block, resident blocks per CU.
    python tools/conv_timeline.py gpurun_out/timing_c64_k3.csv.gz ..."""
import csv
import gzip
import sys

import numpy as np

TICK_US = 1e-2        # wall_clock64() runs at 100 MHz on gfx950


def main():
    for fn in sys.argv[1:]:
        op = gzip.open if fn.endswith('.gz') else open
        rows = list(csv.DictReader(op(fn, 'rt')))
        a = np.array([[int(r[k]) for k in ('start', 'pro', 'main', 'end')] for r in rows], dtype=np.float64) * TICK_US
        span = a[:, 3].max() - a[:, 0].min()
        pro, mn, epi, tot = a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 3] - a[:, 2], a[:, 3] - a[:, 0]
        e1 = ''
        if rows and rows[0].get('e1') not in (None, '', '0'):
            t1 = np.array([int(r['e1']) for r in rows], dtype=np.float64) * TICK_US
            e1 = f' (LDS transposition {(t1 - a[:, 2]).mean():.1f} + row loop {(a[:, 3] - t1).mean():.1f})'
        if rows and rows[0].get('clk') not in (None, '', '0'):
            clk = np.array([int(r['clk']) for r in rows], dtype=np.float64)
            e1 += f'; shader clock {np.median(clk / (tot * 1e-6)) / 1e9:.3f} GHz (s_memtime ticks / wall time per block)'
        print(f'{fn}: {len(a)} blocks, kernel span {span:.0f} us; per block prologue {pro.mean():.1f} us, main loop '
              f'{mn.mean():.1f}, epilogue {epi.mean():.1f}{e1}, total {tot.mean():.1f}; resident blocks per CU {tot.sum() / 256 / span:.2f}')


if __name__ == '__main__':
    main()
