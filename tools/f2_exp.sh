export TMPDIR=/tmp
O=gpurun_out/r4b
mkdir -p $O
for e in 0 2 4 6 8 $((16+256*10)) $((16+256*20)) $((16+256*40)) $((16+256*80)); do echo "== TTSAMD_F2_EXP=$e"; TTSAMD_F2_EXP=$e python3 tools/fused_pair_bench.py --cases 32:3:5,64:3:5,64:11:5,128:3:5 --variants 2,3 --reps 10 2>&1 | grep "C="; done > $O/exp2.txt
cat $O/exp2.txt
