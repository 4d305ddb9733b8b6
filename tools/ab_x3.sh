#!/bin/bash
# same-box A/B of switches inside the split-bf16 bench step:  gpurun -- 'bash tools/ab_x3.sh "TTSAMD_BFO_CHAIN=0"'
run() { echo -n "$1: "; env $1 python3 bench.py --precision bf16x3 --no-cpu-baseline --no-small --no-extra --steps 10 --warmup 3 $EXTRA 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.3f ms/step, frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))"; }
run "TTSAMD_NOP=1"
for e in "$@"; do run "$e"; done
run "TTSAMD_NOP=1"
for e in "$@"; do run "$e"; done
