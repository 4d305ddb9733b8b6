#!/bin/bash
# batch 1 / 8 of the fp32 step with and without the small-problem routing of the fused pair (every pair one launch of 128-column blocks)
O=gpurun_out/r4b
mkdir -p $O
for e in "TTSAMD_FUSED2=0" "TTSAMD_FUSED2_SMALL=0" "TTSAMD_FUSED2_SMALL=1" "TTSAMD_FUSED2_SMALL=1 TTSAMD_HIFIGAN_STREAMS=0"; do
  for b in 1 2 4 8; do echo -n "$e batch $b: "; env $e python3 bench.py --batch $b --no-cpu-baseline --no-small --no-extra --steps 30 --warmup 5 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.3f ms/step, median %.3f' % (d['ms_per_step'], d['ms_per_step_median']))"; done; done | tee $O/f2_small.txt
