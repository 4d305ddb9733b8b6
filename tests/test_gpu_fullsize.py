"""GPU parity at BASELINE's FULL sizes, every utterance (pytest -m gpu): the whole bench.py workload
(config 2: 32 utterances x 64 tokens, forced durations) and config 3's per-GPU share in its bf16 modes.

The checker is oracle/tts_oracle.py — the restatement of the reference pinned to the real reference by
tests/test_oracle_golden.py — run with its tensors ON THE GPU (torch-ROCm: MIOpen / rocBLAS fp32), because on
the host it needs ~1 s per utterance.  It follows the reference's plumbing exactly: FastPitch on the padded
batch, then the vocoder looped per utterance on exact-length mels (models/fastpitch/networks.py:334-345).
Tolerances (BASELINE.json north_star): mel 1e-3, wave 1e-4 max-abs, dec_lens exact; bf16 operands have their own
stated tolerances (tests/test_gpu_parity.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

MEL_TOL, WAVE_TOL = 1e-3, 1e-4                  # fp32 and split-bf16
BF16_MEL_TOL, BF16_WAVE_TOL = 6e-2, 4e-2        # plain bf16 operands (8-bit mantissa) through ~75 convs
B, LT = 32, 64


@pytest.fixture(scope='module')
def workload(synth_weights):
    assert torch.cuda.is_available(), 'gpu tests need an MI355X'
    import tts_oracle as O
    from ttsamd import synth
    from ttsamd.config import NET_CONFIG, HIFIGAN_CONFIG
    dev = torch.device('cuda:0')
    ids_np, dur_np = synth.synth_ids(B, LT), synth.synth_durations(B, LT)
    fw = {k: v.to(dev) for k, v in O.to_torch(synth_weights['fastpitch']).items()}
    hw = {k: v.to(dev) for k, v in O.fold_weight_norm(synth_weights['hifigan']).items()}
    with torch.inference_mode(), torch.device(dev):
        mel_ref, lens_ref, waves_ref = O.tts_batch(fw, NET_CONFIG, hw, HIFIGAN_CONFIG, ids_np,
                                                   dur_tgt=torch.from_numpy(dur_np).to(dev))
    torch.cuda.synchronize()
    # the GPU-resident checker itself against the host oracle on one utterance (the longest: batch-independent)
    lens = np.asarray(lens_ref.cpu())
    bmax = int(np.argmax(lens))
    with torch.inference_mode():
        m1, l1, w1 = O.tts_batch(O.to_torch(synth_weights['fastpitch']), NET_CONFIG,
                                 O.fold_weight_norm(synth_weights['hifigan']), HIFIGAN_CONFIG, ids_np[bmax:bmax + 1],
                                 dur_tgt=dur_np[bmax:bmax + 1])
    assert int(l1[0]) == int(lens[bmax])
    assert float((mel_ref[bmax, :, :int(l1[0])].cpu() - m1[0, :, :int(l1[0])]).abs().max()) < 1e-4
    assert float((waves_ref[bmax].cpu().reshape(-1) - w1[0].reshape(-1)).abs().max()) < 2e-5
    return {'dev': dev, 'ids': torch.from_numpy(ids_np).to(dev), 'dur': torch.from_numpy(dur_np).to(dev),
            'mel': mel_ref.cpu(), 'lens': lens, 'waves': [w.reshape(-1).cpu() for w in waves_ref]}


@pytest.mark.parametrize('mode,mel_tol,wave_tol', [('f32', MEL_TOL, WAVE_TOL), ('bf16x3', MEL_TOL, WAVE_TOL),
                                                   ('bf16', BF16_MEL_TOL, BF16_WAVE_TOL)])
def test_full_batch_every_utterance(workload, synth_weights, mode, mel_tol, wave_tol):
    from ttsamd.engine import FastPitchEngine, HifiGanEngine, set_precision
    set_precision(mode)
    try:
        dev = workload['dev']
        fp, hg = FastPitchEngine(synth_weights['fastpitch'], device=dev), HifiGanEngine(synth_weights['hifigan'], device=dev)
        mel, dec_lens, *_ = fp.infer(workload['ids'], dur_tgt=workload['dur'])
        wave = hg.forward(mel, dec_lens)
        torch.cuda.synchronize()
        dl = dec_lens.cpu().numpy()
        assert np.array_equal(dl, workload['lens'])                                  # exact
        mel, wave = mel.cpu(), wave.cpu()
        worst_mel = worst_wave = 0.0
        for b in range(B):
            n = int(dl[b])
            worst_mel = max(worst_mel, float((mel[b, :, :n] - workload['mel'][b, :, :n]).abs().max()))
            assert workload['waves'][b].numel() == 256 * n
            worst_wave = max(worst_wave, float((wave[b, :256 * n] - workload['waves'][b]).abs().max()))
            assert float(wave[b, 256 * n:].abs().max()) == 0.0 if 256 * n < wave.shape[1] else True
        print(f'full-size {mode}: mel max-abs {worst_mel:.2e} (tol {mel_tol}), wave max-abs {worst_wave:.2e} (tol {wave_tol})')
        assert worst_mel < mel_tol and worst_wave < wave_tol
    finally:
        set_precision('f32')


def test_config5_full_batch_every_utterance(synth_weights):
    """BASELINE config 5 at full size: 4-speaker FastPitch (32 x 64 tokens, forced durations, one speaker id per call as the
    reference API has it: models/fastpitch/fastpitch/model.py:358-359) -> MelVocos('22k') on the ragged batch, every utterance
    against the oracle run with its tensors on the GPU (FastPitch on the padded batch, Vocos per utterance on exact-length mels)."""
    import tts_oracle as O
    from ttsamd import synth
    from ttsamd.config import NET_CONFIG, VOCOS_22K_CONFIG
    from ttsamd.engine import FastPitchEngine, VocosEngine
    dev = torch.device('cuda:0')
    cfg4 = dict(NET_CONFIG, n_speakers=4, speaker_emb_weight=1.0)
    sd4 = synth.fastpitch_state_dict(cfg4)
    vw = synth.vocos_state_dict()
    ids_np, dur_np = synth.synth_ids(B, LT), synth.synth_durations(B, LT)
    fw = {k: v.to(dev) for k, v in O.to_torch(sd4).items()}
    vwd = {k: torch.as_tensor(np.asarray(v)).to(dev) for k, v in vw.items()}
    fp, voc = FastPitchEngine(sd4, cfg4, device=dev), VocosEngine(vw, device=dev)
    worst_mel = worst_wave = 0.0
    for spk in (1, 3):
        with torch.inference_mode(), torch.device(dev):
            mel_ref, lens_ref, *_ = O.fastpitch_infer(fw, cfg4, ids_np, dur_tgt=torch.from_numpy(dur_np).to(dev), speaker=spk)
        mel, dec_lens, *_ = fp.infer(torch.from_numpy(ids_np).to(dev), dur_tgt=torch.from_numpy(dur_np).to(dev), speaker=spk)
        wave = voc.forward(mel, dec_lens)
        torch.cuda.synchronize()
        dl = dec_lens.cpu().numpy()
        assert np.array_equal(dl, np.asarray(lens_ref.cpu()))
        for b in range(B):
            n = int(dl[b])
            worst_mel = max(worst_mel, float((mel[b, :, :n] - mel_ref[b, :, :n]).abs().max()))
            with torch.inference_mode(), torch.device(dev):
                ref = O.vocos_forward(vwd, mel_ref[b:b + 1, :, :n], VOCOS_22K_CONFIG)[0]
            worst_wave = max(worst_wave, float((wave[b, :256 * n] - ref).abs().max()))
            assert n * 256 == wave.shape[1] or float(wave[b, 256 * n:].abs().max()) == 0.0
    print(f'full-size config 5: mel max-abs {worst_mel:.2e} (tol {MEL_TOL}), wave max-abs {worst_wave:.2e} (tol {WAVE_TOL})')
    assert worst_mel < MEL_TOL and worst_wave < WAVE_TOL


def test_config4_full_size_tacotron2_448_steps():
    """BASELINE config 4 at the bench's size: batch 8 x 64 tokens, 448 decoder steps (gate biased shut so every run decodes
    exactly 448 frames, prenet dropout ON with the shared hash masks), persistent decoder (the default) AND the graph path,
    against oracle/taco_oracle.py on the host.  PARITY UNPINNED (torchaudio's Tacotron2 is not in the reference tree, SURVEY
    §8c): this proves HIP == restatement over a 448-step recurrence, not HIP == reference.  Rounding differences feed back
    through the attention / prenet loop: mel 1e-3 (north star) on the whole trajectory, alignments 1e-4 over the first 32
    steps and 1e-2 at the end of the trajectory, lengths exact."""
    import os
    import taco_oracle as T
    from ttsamd.config import TACOTRON2_CONFIG
    from ttsamd.engine import Tacotron2Engine
    from ttsamd.synth import tacotron2_state_dict, synth_ids
    dev = torch.device('cuda:0')
    bt, frames = 8, 448
    cfg = dict(TACOTRON2_CONFIG)
    sd = tacotron2_state_dict(cfg, seed=0, gate_bias=-30.0)
    tok = torch.from_numpy(synth_ids(bt, 64))
    lens = torch.full((bt,), 64, dtype=torch.int64)
    sids = torch.arange(bt) % max(1, cfg.get('num_speakers', 1))
    with torch.inference_mode():
        mel_ref, lens_ref, al_ref = T.tacotron2_infer(sd, cfg, tok, sids, lens, max_step=frames, seed=7)
    eng = Tacotron2Engine(sd, cfg, device=dev)
    for mode in ('2', '0'):                                                    # persistent dataflow decoder, then the hipGraph path
        os.environ['TTSAMD_TACO_PERSISTENT'] = mode
        try:
            mel, mel_lens, al = eng.infer(tok.to(dev), sids.to(dev), lens.to(dev), max_step=frames, dropout_seed=7)
        finally:
            os.environ.pop('TTSAMD_TACO_PERSISTENT', None)
        assert mel.shape == (bt, 80, frames) and mel_lens.cpu().tolist() == np.asarray(lens_ref).tolist() == [frames] * bt
        em = float((mel.cpu() - mel_ref).abs().max())
        ea = float((al.cpu() - al_ref).abs().max())
        ea32 = float((al.cpu()[:, :32] - al_ref[:, :32]).abs().max())
        print(f'full-size config 4 (TTSAMD_TACO_PERSISTENT={mode}): mel max-abs {em:.2e}, alignments max-abs {ea32:.2e} over the first 32 '
              f'steps, {ea:.2e} over all {frames}')
        # the attention weights feed back into themselves (cumulative location term) and into both LSTMs: fp32 rounding
        # differences between two correct implementations grow along the 448-step trajectory (measured 2.8e-3 at the end
        # vs < 1e-4 over the first 32 steps); the mel, which the north star's tolerance is stated on, stays inside 1e-3
        assert em < MEL_TOL and ea32 < 1e-4 and ea < 1e-2
