# Timing experiments on the F(4,3) kernel (conv_wino4.hip, -DTTS_W4_EXP=<bits>: main loop without its x loads (1) / activations (2) /
# plane writes (4) / weight refills (8) / B reads (16)).  The variants give WRONG results by design: they are separate libraries under
# tools/bin/w4exp/<bits>/ (built in the build container: see the loop below) and the binding is pointed at them with TTSAMD_LIB.
#   for V in 7 1 4 31 24; do hipcc ... -DTTS_W4_EXP=$V -c conv_wino4.hip -o /tmp/w4exp_$V.o; hipcc -shared -o tools/bin/w4exp/$V/libttsamd.so <other objects> /tmp/w4exp_$V.o; done
#   gpurun -- 'bash tools/w4_exp.sh'
O=gpurun_out/w4_exp; mkdir -p $O
export WINO_BENCH_ONLY=wino2,wino4
echo "== product library" > $O/exp.txt
python3 tools/wino_bench.py 128 256 2>&1 | grep -v "k=3" >> $O/exp.txt
for V in 1 4 7 24 31; do
  echo "== TTS_W4_EXP=$V" >> $O/exp.txt
  TTSAMD_LIB=tools/bin/w4exp/$V/libttsamd.so WINO_BENCH_ONLY=wino4 python3 tools/wino_bench.py 128 256 2>&1 | grep -v "k=3" >> $O/exp.txt
done
cat $O/exp.txt
