#!/bin/bash
# Same-box A/B of the fp32 fused-pair routing inside the bench step (B = 32 x 64 tokens): TTSAMD_FUSED2_MASK bit 3 ci + ki,
# ci = C 32 / 64 / 128, ki = k 3 / 7 / 11; _MASK_N1: which of them take 128-column blocks.   gpurun -- 'bash tools/f2_ab.sh'
O=gpurun_out/r4b
mkdir -p $O
run() { echo -n "$1: "; env $1 python3 bench.py --no-cpu-baseline --no-small --no-extra --steps 10 $2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.2f ms/step, median %.2f, frac %.4f' % (d['ms_per_step'], d['ms_per_step_median'], d['roofline']['frac']))"; }
{
run "TTSAMD_FUSED2=0"
run "TTSAMD_FUSED2_MASK=007 TTSAMD_FUSED2_MASK_N1=000"
run "TTSAMD_FUSED2_MASK=00f TTSAMD_FUSED2_MASK_N1=000"
run "TTSAMD_FUSED2_MASK=00f TTSAMD_FUSED2_MASK_N1=008"
run "TTSAMD_FUSED2_MASK=01f TTSAMD_FUSED2_MASK_N1=008"
run "TTSAMD_FUSED2_MASK=05f TTSAMD_FUSED2_MASK_N1=048"
run "TTSAMD_FUSED2_MASK=0df TTSAMD_FUSED2_MASK_N1=048"
run "TTSAMD_FUSED2_MASK=03f TTSAMD_FUSED2_MASK_N1=008"
run "TTSAMD_FUSED2_MASK=05f TTSAMD_FUSED2_MASK_N1=04f"
run "TTSAMD_FUSED2=0" --pipeline
run "TTSAMD_FUSED2_MASK=05f TTSAMD_FUSED2_MASK_N1=048" --pipeline
} | tee $O/f2_ab.txt
