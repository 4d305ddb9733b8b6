#!/bin/bash
# bf16 step with the k = 7 ResBlocks of the C = 32 / 64 stages as one chained launch each (TTSAMD_BFO_CHAIN7=1) or as three pair launches (=0)
run() { echo -n "$1 $2: "; env $1 python3 bench.py --precision bf16 --no-cpu-baseline --no-small --no-extra --steps 30 --warmup 5 $2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.3f ms/step' % d['ms_per_step'])"; }
for rep in 1 2; do for e in 0 1; do run TTSAMD_BFO_CHAIN7=$e --no-pipeline; run TTSAMD_BFO_CHAIN7=$e ""; done; done
for e in 0 1; do run TTSAMD_BFO_CHAIN7=$e "--no-pipeline --batch 8"; run TTSAMD_BFO_CHAIN7=$e "--no-pipeline --batch 1"; done
