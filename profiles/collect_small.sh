#!/bin/bash
# Kernel traces of the small-batch cases of the headline path (BASELINE target "batch 1/8/32"):
#   gpurun --timeout 900 -- 'bash profiles/collect_small.sh r1'
set -u
R=${1:-r1}
O=gpurun_out/small_$R
mkdir -p $O
export TMPDIR=/tmp
for b in 1 8; do
  python3 bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline > $O/b${b}_bench_line.json 2>> $O/err.log
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b$b -- python3 bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline > $O/stats_b$b.log 2>&1
  python3 profiles/summarize.py $(ls $O/stats_b$b/*/*kernel_trace.csv | head -1) > $O/b${b}_by_grid.txt
  cp $(ls $O/stats_b$b/*/*kernel_stats.csv | head -1) $O/b${b}_kernel_stats.csv
  rm -rf $O/stats_b$b
done
cat $O/b1_bench_line.json $O/b8_bench_line.json | cut -c1-400
