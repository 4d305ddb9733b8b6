"""One HiFi-GAN forward at batch 1 (448 frames) repeated 5 times: run under rocprofv3 --kernel-trace to get the launch timeline.
gpurun -- 'rocprofv3 --kernel-trace --output-format csv -d gpurun_out/hgtrace -- python3 tools/hg_b1_trace.py; python3 tools/hg_b1_timeline.py gpurun_out/hgtrace'"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tts-arabic-pytorch_amd'))
from ttsamd import synth  # noqa: E402
from ttsamd.engine import HifiGanEngine  # noqa: E402

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
hg = HifiGanEngine(synth.hifigan_state_dict(), device=dev)
mel = torch.randn(B, 80, 448, device=dev)
lens = torch.full((B,), 448, device=dev)
for _ in range(5):
    hg.forward(mel, lens)
    torch.cuda.synchronize()
