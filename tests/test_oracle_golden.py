"""CPU: pin oracle/tts_oracle.py against outputs of the real reference (tests/golden/,
made by oracle/gen_golden.py).  Tolerances: the oracle issues the same ATen ops as the
reference, so agreement is expected to ~1e-6; indices are exact."""
import numpy as np
import pytest
import torch

import tts_oracle as O
from ttsamd.config import NET_CONFIG, HIFIGAN_CONFIG


def maxabs(a, b):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0.0


def test_weightnorm_fold(golden):
    g = golden('weightnorm_fold')
    sd = {'c.parametrizations.weight.original0': g['conv_g'], 'c.parametrizations.weight.original1': g['conv_v'],
          't.weight_g': g['convt_g'], 't.weight_v': g['convt_v']}
    w = O.fold_weight_norm(sd)
    assert maxabs(w['c.weight'], g['conv_w']) == 0.0
    assert maxabs(w['t.weight'], g['convt_w']) == 0.0


@pytest.mark.parametrize('T', [1, 7, 40])
def test_hifigan(golden, synth_weights, T):
    g = golden(f'hifigan_T{T}')
    w = O.fold_weight_norm(synth_weights['hifigan'])
    stages = []
    wave = O.hifigan_forward(w, g['mel'], HIFIGAN_CONFIG, stages=stages)
    assert wave.shape == (1, 256 * T)
    assert maxabs(wave, g['wave']) < 2e-6
    if T == 7:
        assert maxabs(stages[0], g['stage_conv_pre'][None] if g['stage_conv_pre'].ndim == 2 else g['stage_conv_pre']) < 1e-5
        for i in range(4):
            ref = g[f'stage_ups{i}']
            ref = ref[None] if ref.ndim == 2 else ref
            assert maxabs(stages[1 + 2 * i], ref) < 1e-5
    # 3-D input (test.py:62-63 calls vocoder(mel[None]))
    wave3 = O.hifigan_forward(w, g['mel'][None], HIFIGAN_CONFIG)
    assert maxabs(wave3[0], g['wave']) < 2e-6


def test_regulate_len_indices_exact(golden):
    g = golden('regulate_len')
    for tag, pace in (('0p8', 0.8), ('1p0', 1.0), ('1p25', 1.25)):
        reps, dec_lens, idx = O.regulate_len_indices(g['dur'], pace)
        assert np.array_equal(dec_lens, g[f'dec_lens_{tag}'])
        assert np.array_equal(idx, g[f'idx_{tag}'])
        # gather formulation == the reference's dense one-hot matmul, bit for bit
        enc = g['enc']
        rep1 = np.where(idx >= 0, np.take_along_axis(enc[:, :, 1], np.maximum(idx, 0), axis=1), 0.0)
        assert np.array_equal(rep1.astype(np.float32), g[f'rep1_{tag}'])


def test_fastpitch_ragged_batch(golden, synth_weights):
    g = golden('fastpitch_b3_durtgt')
    w = O.to_torch(synth_weights['fastpitch'])
    trace = {}
    mel, dec_lens, dur, pitch, energy = O.fastpitch_infer(w, NET_CONFIG, g['ids'], dur_tgt=g['dur_tgt'], trace=trace)
    assert np.array_equal(dec_lens.numpy(), g['dec_lens'])
    assert maxabs(trace['encoder.layers.0.out'], g['encoder_l0_ff'] * (g['ids'] != 0)[:, :, None]) < 1e-5
    assert maxabs(trace['enc_out'], g['encoder_out']) < 1e-5
    assert maxabs(dur, g['dur_pred']) < 1e-4
    assert maxabs(pitch, g['pitch_pred']) < 1e-5
    assert maxabs(energy, g['energy_pred']) < 1e-5
    assert maxabs(mel, g['mel']) < 2e-5


@pytest.mark.parametrize('tag', ['p1', 'p0p9_pitch'])
def test_fastpitch_predicted_durations(golden, synth_weights, tag):
    g = golden(f'fastpitch_b2_pred_{tag}')
    w = O.to_torch(synth_weights['fastpitch'])
    mul, add = float(g['pitch_mul']), float(g['pitch_add'])
    ptr = None
    if mul != 1.0 or add != 0.0:
        ptr = lambda p, n, mean, std: mul * p + add          # models/fastpitch/networks.py:38-42
    mel, dec_lens, dur, pitch, energy = O.fastpitch_infer(w, NET_CONFIG, g['ids'], pace=float(g['pace']),
                                                           pitch_transform=ptr)
    assert np.array_equal(dec_lens.numpy(), g['dec_lens'])
    assert maxabs(dur, g['dur_pred']) < 1e-4
    assert maxabs(pitch, g['pitch_pred']) < 1e-5
    assert maxabs(mel, g['mel']) < 2e-5


def test_fastpitch_multispeaker(golden, synth_weights):
    g = golden('fastpitch_b3_spk2')
    w = O.to_torch(synth_weights['fastpitch_spk4'])
    cfg = dict(NET_CONFIG, n_speakers=4)
    mel, dec_lens, *_ = O.fastpitch_infer(w, cfg, g['ids'], dur_tgt=g['dur_tgt'], speaker=2)
    assert np.array_equal(dec_lens.numpy(), g['dec_lens'])
    assert maxabs(mel, g['mel']) < 2e-5


def test_end_to_end_tts(golden, synth_weights):
    """FastPitch2Wave.tts(list, batch_size=3, denoise=0) and (…, denoise=0.005) on three
    infer_text.txt lines; ids come from the committed tokenisation fixture."""
    e = golden('e2e_tts')
    t = golden('infer_text_ids')
    fw = O.to_torch(synth_weights['fastpitch'])
    hw = O.fold_weight_norm(synth_weights['hifigan'])
    seqs = [t['flat'][t['offsets'][i]:t['offsets'][i + 1]] for i in e['line_idx']]
    # text_collate_fn (networks.py:16-35): sort by length desc, zero-pad
    order = np.argsort([-len(s) for s in seqs], kind='stable')
    ids = np.zeros((3, max(map(len, seqs))), np.int64)
    for r, i in enumerate(order):
        ids[r, :len(seqs[i])] = seqs[i]
    mel, dec_lens, waves = O.tts_batch(fw, NET_CONFIG, hw, HIFIGAN_CONFIG, ids)
    for r, i in enumerate(order):
        ref = e[f'wave{i}']
        assert waves[r].shape == ref.shape
        assert maxabs(waves[r], ref) < 1e-5
    # single path (tts_single, return_mel) on the first picked line
    mel1, dl1, w1 = O.tts_batch(fw, NET_CONFIG, hw, HIFIGAN_CONFIG, seqs[0][None])
    assert maxabs(mel1[0], e['single_mel']) < 2e-5
    assert maxabs(w1[0], e['single_wave']) < 1e-5
    # denoiser (torch.stft stand-in on both sides: pins our restatement of denoiser.py:50-72 only)
    bias = O.denoiser_bias_spec(hw, HIFIGAN_CONFIG)
    assert maxabs(bias, e['bias_spec']) < 1e-5
    seqs2 = seqs[:2]
    order2 = np.argsort([-len(s) for s in seqs2], kind='stable')
    ids2 = np.zeros((2, max(map(len, seqs2))), np.int64)
    for r, i in enumerate(order2):
        ids2[r, :len(seqs2[i])] = seqs2[i]
    _, _, wd = O.tts_batch(fw, NET_CONFIG, hw, HIFIGAN_CONFIG, ids2, denoise_strength=0.005, bias_spec=bias)
    for r, i in enumerate(order2):
        assert maxabs(wd[r], e[f'wave_dn{i}']) < 1e-5


def test_vocos_22k(golden):
    """MelVocos('22k') forward (+ denoising vector) of the reference vs the oracle."""
    from ttsamd import synth
    from ttsamd.config import VOCOS_22K_CONFIG
    import hashlib, json, os
    from conftest import GOLDEN
    w = synth.vocos_state_dict()
    h = hashlib.sha256()
    for k in sorted(w):
        h.update(k.encode()); h.update(np.ascontiguousarray(w[k]).tobytes())
    with open(os.path.join(GOLDEN, 'digests.json')) as f:
        assert h.hexdigest() == json.load(f)['vocos_seed0']
    g = golden('vocos_22k')
    bias = O.vocos_bias_vec(w, VOCOS_22K_CONFIG)
    assert maxabs(bias, g['bias_vec']) < 1e-6
    for T in (1, 5, 33):
        assert maxabs(O.vocos_forward(w, g[f'mel_T{T}'], VOCOS_22K_CONFIG, bias_vec=bias), g[f'wave_T{T}']) < 1e-6
        assert maxabs(O.vocos_forward(w, g[f'mel_T{T}'], VOCOS_22K_CONFIG, denoise=0.3, bias_vec=bias), g[f'wave_dn_T{T}']) < 1e-6


def test_ragged_batched_vocoder_oracle_equals_the_per_utterance_loop(synth_weights):
    """`hifigan_forward_ragged` (one padded batch, zero at and past each utterance's own edge before every conv) against the
    reference's plumbing, `hifigan_forward` on each exact-length mel: same sums of the same products at every valid sample.  The
    full-size GPU checks use the batched form (one shape per layer instead of one per distinct length)."""
    import torch
    import tts_oracle as O
    from ttsamd.config import HIFIGAN_CONFIG
    w = O.fold_weight_norm(synth_weights['hifigan'])
    rng = np.random.default_rng(5)
    lens = [9, 1, 4, 7]
    mel = (rng.standard_normal((4, 80, 9)) * 1.5 - 4.0).astype(np.float32)
    mel[1, :, 1:] = 123.0                       # garbage past an utterance's end must not reach its samples
    with torch.inference_mode():
        got = O.hifigan_forward_ragged(w, mel, lens, HIFIGAN_CONFIG)
        for b, n in enumerate(lens):
            ref = O.hifigan_forward(w, mel[b, :, :n], HIFIGAN_CONFIG)[0]
            assert float((got[b, :256 * n] - ref).abs().max()) < 2e-6, b
            assert n == 9 or float(got[b, 256 * n:].abs().max()) == 0.0
