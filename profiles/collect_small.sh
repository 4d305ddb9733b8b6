#!/bin/bash
# Kernel tables of the small-batch calls of the headline path (BASELINE target "batch 1/8/32"), one-stream schedule, fp32 and split bf16:
#   gpurun --timeout 1200 -- 'bash profiles/collect_small.sh r6'
# per (precision, batch): the bench line (no per-launch events), the per-instantiation kernel table, the kernel stats, GPU idle gaps
set -u
R=${1:-r6}
O=gpurun_out/small_$R
mkdir -p $O
export TMPDIR=/tmp
for p in f32 bf16x3; do
  for b in 1 8; do
    python3 bench.py --precision $p --batch $b --steps 50 --warmup 5 --no-pipeline --no-cpu-baseline --no-small --no-extra > $O/b${b}_${p}_bench_line.json 2>> $O/err.log
    TTSAMD_HIFIGAN_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_${p}_$b -- python3 bench.py --precision $p --batch $b --steps 10 --warmup 3 --no-pipeline --no-cpu-baseline --no-small --no-extra > $O/st_${p}_$b.log 2>&1
    f=$(find $O/st_${p}_$b -name '*kernel_trace.csv' | head -1)
    python3 profiles/summarize.py $f > $O/b${b}_${p}_by_grid_one_stream.txt
    python3 profiles/gaps.py $f > $O/b${b}_${p}_gaps.txt 2>&1
    cp $(find $O/st_${p}_$b -name '*kernel_stats.csv' | head -1) $O/b${b}_${p}_kernel_stats_one_stream.csv
    rm -rf $O/st_${p}_$b
  done
done
for f in $O/b*_bench_line.json; do python3 -c "import json,sys; d=json.loads(open('$f').readline()); print('$f', round(d['ms_per_step'],3), 'ms', round(d['roofline']['frac'],3))"; done
