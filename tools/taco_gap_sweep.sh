#!/bin/bash
# Persistent Tacotron2 decoder: us per step vs the stagger between the two polls in flight at every hand-off
# (-DTACO_POLL_GAP=<n> s_sleep units of 64 clocks).  Run on the GPU box from the repo root.
cd tts-arabic-pytorch_amd/csrc || exit 1
cp ../ttsamd/lib/libttsamd.so /tmp/keep.so
for g in ${1:-0 8 20 40}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DTACO_POLL_GAP=$g -c tacotron2.hip -o /tmp/tg.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../ttsamd/lib/libttsamd.so $(ls build/*.o | grep -v tacotron2.o) /tmp/tg.o -ldl
  echo "gap $g: $(cd ../.. && TTSAMD_TACO_DEBUG=1 timeout 200 python3 tools/taco_bench.py --steps 3 2>&1 | grep 'persistent decoder' | tail -1)"
done
cp /tmp/keep.so ../ttsamd/lib/libttsamd.so
