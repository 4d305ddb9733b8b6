#!/bin/bash
# bf16 small-batch defaults re-checked on one box: every schedule switch of the octet path off / on at batch 1 and 8 (one stream, per-launch events on)
run() { echo -n "$1 batch $2: "; env $1 python3 bench.py --precision bf16 --no-pipeline --no-cpu-baseline --no-small --no-extra --steps 60 --warmup 5 --batch $2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.3f' % d['ms_per_step'])"; }
for b in 1 8; do
  run DEFAULT=1 $b
  for sw in TTSAMD_BFO_SPLITK TTSAMD_BFO_CHAIN TTSAMD_BFO_CHAIN7 TTSAMD_BFO_FUSED_LN TTSAMD_BFO_FF TTSAMD_BF16_ATTN TTSAMD_BFO_SMALL_TILES; do for v in 0 1; do run $sw=$v $b; done; done
  run DEFAULT=1 $b
done
