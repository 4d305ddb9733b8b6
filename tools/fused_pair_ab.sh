#!/bin/bash
# A/B of resblock_pair<K, C> build variants on the GPU box: average duration of the kernel in a 5-step bench under rocprofv3
# (one-stream schedule).  usage: bash tools/fused_pair_ab.sh "<hipcc -D flags variant 1>" "<variant 2>" ...
cd tts-arabic-pytorch_amd/csrc || exit 1
export TMPDIR=/tmp
cp ../ttsamd/lib/libttsamd.so /tmp/libttsamd.keep
for flags in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c resblock_fused.hip -o /tmp/rf.o 2>/dev/null || { echo "build failed: $flags"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../ttsamd/lib/libttsamd.so $(ls build/*.o | grep -v resblock_fused.o) /tmp/rf.o -ldl
  rm -rf /tmp/ab_stats
  (cd ../.. && TTSAMD_HIFIGAN_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-small > /tmp/ab.log 2>&1)
  echo "== $flags"
  f=$(ls /tmp/ab_stats/*/*kernel_stats.csv 2>/dev/null | head -1)
  if [ -z "$f" ]; then tail -5 /tmp/ab.log; else python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'resblock_pair' in r['Name']: print(r['Name'].split('(')[0], r['Calls'], 'avg_us', round(float(r['AverageNs'])/1e3,1))"; fi
done
cp /tmp/libttsamd.keep ../ttsamd/lib/libttsamd.so
