"""Measurement: the reference's algorithm executed by PyTorch-ROCm
(MIOpen / rocBLAS) on the same MI355X, next to the HIP path, on the bench.py workload (32 utterances x 64
tokens, fp32, forced durations).  The torch side is oracle/tts_oracle.py (the restatement of the reference
pinned by tests/test_oracle_golden.py) with its tensors on the GPU; /root/reference itself does not exist on
the GPU box.  Two variants: the reference's own plumbing (batched FastPitch, vocoder looped per utterance,
models/fastpitch/networks.py:340-345) and a batched vocoder call on the padded mel, which is what a user
tuning the reference for throughput would do.  A MEASUREMENT, not a parity test: it costs three minutes (MIOpen compiles a kernel
per distinct utterance length on a fresh box), so it runs only with TTSAMD_TORCH_GPU_BASELINE=1 -- profiles/collect.sh sets it and
keeps gpurun_out/torch_rocm_baseline.json with the round's evidence; the same env var also turns on the timing assertion and the
MIOpen find-mode variant."""
import json
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(os.environ.get('TTSAMD_TORCH_GPU_BASELINE') != '1', reason='measurement, opt-in: TTSAMD_TORCH_GPU_BASELINE=1')
def test_hip_path_beats_pytorch_rocm_on_the_same_gpu(synth_weights):
    import tts_oracle as O
    from ttsamd import synth
    from ttsamd.config import NET_CONFIG, HIFIGAN_CONFIG
    from ttsamd.engine import FastPitchEngine, HifiGanEngine
    dev = torch.device('cuda:0')
    B, Lt = int(os.environ.get('TTSAMD_BASELINE_BATCH', '32')), 64
    ids_np, dur_np = synth.synth_ids(B, Lt), synth.synth_durations(B, Lt)
    ids, dur = torch.from_numpy(ids_np).to(dev), torch.from_numpy(dur_np).to(dev)

    def timed(fn, warm, n):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n, out

    fp, hg = FastPitchEngine(synth_weights['fastpitch'], device=dev), HifiGanEngine(synth_weights['hifigan'], device=dev)

    def ours():
        mel, dec_lens, *_ = fp.infer(ids, dur_tgt=dur)
        return hg.forward(mel, dec_lens), dec_lens

    t_ours, (wave, dec_lens) = timed(ours, 2, 5)
    n_samples = int(dec_lens.sum().item()) * 256

    fw = {k: v.to(dev) for k, v in O.to_torch(synth_weights['fastpitch']).items()}
    hw = {k: v.to(dev) for k, v in O.fold_weight_norm(synth_weights['hifigan']).items()}
    res = {'workload': f'{B} x {Lt} tokens, fp32, forced durations', 'samples_per_step': n_samples,
           'hip_ms': t_ours * 1e3, 'hip_samples_per_s': n_samples / t_ours}
    with torch.inference_mode(), torch.device(dev):
        def torch_reference_plumbing():
            return O.tts_batch(fw, NET_CONFIG, hw, HIFIGAN_CONFIG, ids_np, dur_tgt=dur)

        def torch_batched_vocoder():
            mel, dl, *_ = O.fastpitch_infer(fw, NET_CONFIG, ids_np, dur_tgt=dur)
            return O.hifigan_forward(hw, mel, HIFIGAN_CONFIG), dl

        strict = os.environ.get('TTSAMD_TORCH_GPU_BASELINE') == '1'
        for bench_mode in ((False, True) if strict else (False,)):
            torch.backends.cudnn.benchmark = bench_mode          # MIOpen find mode on / off
            tag = 'find' if bench_mode else 'default'
            t0 = time.perf_counter()
            t_b, (w_b, _) = timed(torch_batched_vocoder, 2, 3)
            res[f'torch_rocm_batched_{tag}_ms'] = t_b * 1e3
            res[f'torch_rocm_batched_{tag}_samples_per_s'] = n_samples / t_b
            res[f'torch_rocm_batched_{tag}_setup_s'] = time.perf_counter() - t0 - 3 * t_b
        torch.backends.cudnn.benchmark = False
        t_l, (_, _, waves) = timed(torch_reference_plumbing, 1, 2)
        res['torch_rocm_reference_plumbing_ms'] = t_l * 1e3
        res['torch_rocm_reference_plumbing_samples_per_s'] = n_samples / t_l
    # same numbers on both sides (the padded batch differs from the per-utterance loop only past each end)
    n0 = int(dec_lens[0]) * 256
    assert float((wave[0, :n0] - waves[0].reshape(-1)[:n0]).abs().max()) < 1e-4
    best_torch = min(v for k, v in res.items() if k.startswith('torch_rocm') and k.endswith('_ms'))
    res['speedup_vs_best_torch_rocm'] = best_torch / res['hip_ms']
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/torch_rocm_baseline.json', 'w') as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))
    if os.environ.get('TTSAMD_TORCH_GPU_BASELINE') == '1':
        assert res['speedup_vs_best_torch_rocm'] > 1.0
