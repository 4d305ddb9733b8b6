#!/bin/bash
# Persistent Tacotron2 decoder: time per step with phases compiled out (-DTP_SKIP bit mask: 1 attention LSTM, 2 query+energies,
# 4 softmax+context, 8 decoder LSTM, 16 projection+prenet-1; results are wrong, the timing shows what each phase costs).
# Run on the GPU box from the repo root: bash tools/taco_phase_timing.sh "0 1 2 4 8 16 31"
cd tts-arabic-pytorch_amd/csrc || exit 1
cp ../ttsamd/lib/libttsamd.so /tmp/libttsamd.keep
for k in ${1:-0 31}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DTP_SKIP=$k -c tacotron2.hip -o /tmp/taco_skip.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../ttsamd/lib/libttsamd.so $(ls build/*.o | grep -v tacotron2.o) /tmp/taco_skip.o -ldl
  echo "TP_SKIP=$k: $(cd ../.. && TTSAMD_TACO_DEBUG=1 timeout 300 python3 tools/taco_bench.py --steps 3 2>&1 | grep 'persistent decoder' | tail -1)"
done
if [ -n "$2" ]; then   # second argument: also the per-phase stamps of one step
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DTP_TIMING -c tacotron2.hip -o /tmp/taco_skip.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../ttsamd/lib/libttsamd.so $(ls build/*.o | grep -v tacotron2.o) /tmp/taco_skip.o -ldl
  (cd ../.. && timeout 300 python3 tools/taco_phase_stamps.py 2>&1 | tail -14; timeout 300 python3 tools/taco_flow_stamps.py 2>&1 | tail -40)
fi
cp /tmp/libttsamd.keep ../ttsamd/lib/libttsamd.so
