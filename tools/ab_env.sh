#!/bin/bash
# same-box A/B of schedule / block-order switches inside the fp32 bench step:  gpurun -- 'bash tools/ab_env.sh "TTSAMD_XCD_W=0" "TTSAMD_COMPACT=0"'
run() { echo -n "$1: "; env $1 python3 bench.py --no-cpu-baseline --no-small --no-extra --steps 20 --warmup 4 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.2f ms/step, frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))"; }
run "TTSAMD_NOP=1"
for e in "$@"; do run "$e"; done
run "TTSAMD_NOP=1"
