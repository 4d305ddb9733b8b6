#!/bin/bash
# kernel tables of the batch-1 and batch-8 fp32 calls (one-stream and default schedule) + GPU idle gaps: where the small-batch time goes
O=gpurun_out/small; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for b in 1 8; do
  for s in 0 1; do
    TTSAMD_HIFIGAN_STREAMS=$s rocprofv3 --kernel-trace --output-format csv -d $O/b${b}_s$s -- python3 bench.py --batch $b --steps 20 --warmup 3 --no-pipeline --no-cpu-baseline --no-small --no-extra > $O/b${b}_s$s.log 2>&1
    f=$(find $O/b${b}_s$s -name '*kernel_trace.csv' | head -1)
    python3 profiles/summarize.py $f > $O/b${b}_s${s}_by_grid.txt
    python3 profiles/gaps.py $f > $O/b${b}_s${s}_gaps.txt 2>&1
    tail -1 $O/b${b}_s$s.log | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('batch $b streams $s', d['ms_per_step'])"
  done
done
