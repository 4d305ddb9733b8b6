#!/bin/bash
# fp32 step with the co-tiles of one time tile on ONE XCD as long as the XCD's weight slice stays under TTSAMD_XCD_WMAX_KB (0 = one co-tile class per XCD)
run() { echo -n "$1 $2: "; env $1 python3 bench.py --no-cpu-baseline --no-small --no-extra --steps ${STEPS:-12} --warmup 3 $2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.3f ms/step  frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))"; }
for rep in 1 2; do for w in ${WMAX:-0 800 1500 2000 3000}; do run TTSAMD_XCD_WMAX_KB=$w "--no-pipeline"; done; done
for w in ${WMAX2:-0 800 3000}; do run TTSAMD_XCD_WMAX_KB=$w ""; done
