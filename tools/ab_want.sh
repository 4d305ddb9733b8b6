#!/bin/bash
# tile choice of the fp32 conv engine at mid-size batches: TTSAMD_WANT_BLOCKS = blocks a launch should have before a larger tile is taken (default 768)
for b in ${BATCHES:-8 16}; do for w in ${WANTS:-768 1200 2000 3000 4500}; do echo -n "batch $b want $w: "; TTSAMD_WANT_BLOCKS=$w python3 bench.py --no-pipeline --no-cpu-baseline --no-small --no-extra --steps 20 --warmup 3 --batch $b 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"; done; done
