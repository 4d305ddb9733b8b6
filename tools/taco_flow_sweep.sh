#!/bin/bash
# Persistent Tacotron2 decoder, dataflow schedule (TTSAMD_TACO_PERSISTENT=2): us per step vs the back-off between polls
# (-DTACO_POLL_SLEEP=<n>, s_sleep units of 64 clocks).  Run on the GPU box from the repo root.
cd tts-arabic-pytorch_amd/csrc || exit 1
cp ../ttsamd/lib/libttsamd.so /tmp/libttsamd.keep
for k in ${1:-0 8 32}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DTACO_POLL_SLEEP=$k -c tacotron2.hip -o /tmp/taco_sw.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../ttsamd/lib/libttsamd.so $(ls build/*.o | grep -v tacotron2.o) /tmp/taco_sw.o -ldl
  echo "TACO_POLL_SLEEP=$k: $(cd ../.. && TTSAMD_TACO_PERSISTENT=2 TTSAMD_TACO_DEBUG=1 timeout 300 python3 tools/taco_bench.py --steps 3 2>&1 | grep 'persistent decoder' | tail -1)"
done
cp /tmp/libttsamd.keep ../ttsamd/lib/libttsamd.so
