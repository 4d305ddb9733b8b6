"""GPU parity (pytest -m gpu): the HIP path, called through the C ABI, against the CPU
oracle on the same seeded inputs and against the committed reference goldens.
Tolerances (BASELINE.json north_star): mel 1e-3 max-abs, waveform 1e-4 max-abs,
length-regulator indices bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

MEL_TOL = 1e-3
WAVE_TOL = 1e-4


def maxabs(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0.0


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'gpu tests need an MI355X'
    from ttsamd import lib
    assert lib.load().ttsamd_device_ok() == 1
    return torch.device('cuda:0')


@pytest.mark.parametrize('cin,cout,k,dil,lin,B', [
    (32, 32, 3, 1, 300, 2), (32, 32, 11, 5, 1000, 2), (64, 64, 7, 3, 777, 3), (128, 128, 11, 5, 515, 2),
    (256, 256, 3, 1, 130, 2), (80, 512, 7, 1, 40, 2), (384, 1536, 3, 1, 64, 3), (1536, 384, 3, 1, 100, 2),
    (384, 192, 1, 1, 64, 2), (64, 384, 1, 1, 33, 2), (384, 80, 1, 1, 200, 2), (384, 256, 3, 1, 16, 3),
    (16, 32, 3, 1, 1, 1),
])
def test_conv1d_kernel(dev, cin, cout, k, dil, lin, B):
    """One MFMA conv launch vs torch's fp32 conv on the CPU; ragged lengths, leaky-relu on load."""
    from ttsamd.engine import conv1d
    g = torch.Generator().manual_seed(cin * 1000 + cout + k)
    x = torch.randn(B, cin, lin, generator=g)
    w = torch.randn(cout, cin, k, generator=g) / np.sqrt(cin * k)
    b = torch.randn(cout, generator=g)
    lens = torch.tensor([lin, max(1, lin - 7), max(1, lin // 2)][:B], dtype=torch.int64)
    y = conv1d(x.to(dev), w.to(dev), b.to(dev), lens.to(dev), dilation=dil, in_slope=0.1).cpu()
    for i in range(B):
        n = int(lens[i])
        ref = torch.nn.functional.conv1d(torch.nn.functional.leaky_relu(x[i:i + 1, :, :n], 0.1), w, b,
                                         dilation=dil, padding=(k * dil - dil) // 2)[0]
        assert maxabs(y[i, :, :n], ref) < 2e-5, (i, n)
        assert float(y[i, :, n:].abs().max()) == 0.0 if n < lin else True


def _conv_fuzz(dev, cases, tol, seed):
    from ttsamd.engine import conv1d
    rng = np.random.default_rng(seed)
    for (cin, cout, k, dil, lin, B) in cases:
        g = torch.Generator().manual_seed(cin + 7 * cout + 13 * k + lin)
        x = torch.randn(B, cin, lin, generator=g)
        w = torch.randn(cout, cin, k, generator=g) / np.sqrt(cin * k)
        b = torch.randn(cout, generator=g)
        lens = torch.from_numpy(rng.integers(lin // 2, lin + 1, size=B)).long()
        lens[0], lens[B - 1] = lin, 0
        if B > 4:
            lens[1], lens[2], lens[3] = 1, 255, 257
        y = conv1d(x.to(dev), w.to(dev), b.to(dev), lens.to(dev), dilation=dil, in_slope=0.1).cpu()
        for i in rng.choice(B, size=min(B, 6), replace=False).tolist() + [0, B - 1]:
            n = int(lens[i])
            if n:
                ref = torch.nn.functional.conv1d(torch.nn.functional.leaky_relu(x[i:i + 1, :, :n].double(), 0.1),
                                                 w.double(), b.double(), dilation=dil, padding=(k * dil - dil) // 2)[0]
                assert maxabs(y[i, :, :n], ref) < tol, (cin, cout, k, dil, lin, i, n)
            assert float(y[i, :, n:].abs().max()) == 0.0 if n < lin else True, (cin, cout, k, dil, lin, i, n)


def _large_tile_cases(seed, ks=(3, 7, 11)):
    rng = np.random.default_rng(seed)
    cases = []
    for c in (32, 64, 128, 256):
        for k in ks:
            big = int(rng.choice([2048, 4096 + 1, 6144 - 1]))
            cases.append((c, c, k, int(rng.integers(1, 6)), big, 24 if c < 256 else 12))
    return cases + [(384, 1536, 3, 1, 500, 16), (1536, 384, 3, 1, 497, 16), (80, 512, 7, 1, 513, 16), (512, 80, 5, 2, 300, 8)]


def test_conv1d_kernel_large_tiles_fuzz(dev):
    """The cases above are small, so the launcher always picks its smallest tiles.  Here batch x length is
    large enough for every tile shape of the launcher (128x128, 64x256, 32x256, 128x64, 64x64, 32x128), with
    random dilations, lengths straddling tile edges and ragged utterances (some empty), against torch on the
    CPU in float64."""
    _conv_fuzz(dev, _large_tile_cases(2024), 3e-5, 2024)


@pytest.mark.gpu
def test_conv1d_kernel_ragged_batch_over_64_utterances(dev):
    """Ragged launches look their (utterance, tile) pair up in the list of LIVE tiles, 64 utterances per pass
    (conv_mfma_common.hpp: live_tile): more than 64 utterances, some empty, lengths on and around tile edges."""
    _conv_fuzz(dev, [(32, 32, 3, 1, 700, 70), (64, 64, 7, 3, 520, 130), (128, 128, 3, 2, 300, 67)], 3e-5, 77)


def test_length_regulate_exact(dev, golden):
    import tts_oracle as O
    from ttsamd.engine import length_regulate
    g = golden('regulate_len')
    enc = torch.from_numpy(g['enc']).permute(0, 2, 1).contiguous()          # channel-first
    for tag, pace in (('0p8', 0.8), ('1p0', 1.0), ('1p25', 1.25)):
        reps, dec_lens, idx_ref = O.regulate_len_indices(g['dur'], pace)
        out, idx = length_regulate(enc.to(dev), torch.from_numpy(reps).to(dev), idx_ref.shape[1])
        assert np.array_equal(idx.cpu().numpy(), idx_ref)
        assert np.array_equal(idx.cpu().numpy(), g[f'idx_{tag}'])
        assert np.array_equal(out[:, 1].cpu().numpy(), g[f'rep1_{tag}'])     # bit-exact gather


def test_length_regulate_large_property(dev):
    """Full-size property: sorted indices, each token repeated exactly reps times."""
    from ttsamd.engine import length_regulate
    from ttsamd import synth
    B, Lt = 32, 64
    dur = synth.synth_durations(B, Lt)
    reps = torch.from_numpy(dur).long()
    T = int(reps.sum(1).max())
    enc = torch.arange(Lt, dtype=torch.float32)[None, None, :].repeat(B, 2, 1)
    out, idx = length_regulate(enc.to(dev), reps.to(dev), T)
    idx = idx.cpu().numpy()
    for b in range(B):
        n = int(reps[b].sum())
        assert np.all(np.diff(idx[b, :n]) >= 0) and np.all(idx[b, n:] == -1)
        assert np.array_equal(np.bincount(idx[b, :n], minlength=Lt), reps[b].numpy())
        assert np.array_equal(out[b, 0, :n].cpu().numpy(), idx[b, :n].astype(np.float32))


@pytest.fixture(scope='module')
def hifigan_engine(dev, synth_weights):
    from ttsamd.engine import HifiGanEngine
    return HifiGanEngine(synth_weights['hifigan'])


@pytest.mark.parametrize('T', [1, 7, 40])
def test_hifigan_golden(dev, golden, hifigan_engine, T):
    g = golden(f'hifigan_T{T}')
    wave = hifigan_engine.forward(torch.from_numpy(g['mel'])[None].to(dev))
    assert wave.shape == (1, 256 * T)
    assert maxabs(wave[0], g['wave'][0]) < WAVE_TOL


def test_hifigan_ragged_batch_matches_unbatched(dev, synth_weights, hifigan_engine):
    """Batched ragged vocoder == the reference's per-utterance loop (networks.py:340-341):
    every layer must zero-pad at the true utterance edge."""
    import tts_oracle as O
    from ttsamd.config import HIFIGAN_CONFIG
    w = O.fold_weight_norm(synth_weights['hifigan'])
    rng = np.random.default_rng(3)
    lens = [23, 9, 16, 1]
    mel = (rng.standard_normal((4, 80, 23)) * 1.5 - 4.0).astype(np.float32)
    wave = hifigan_engine.forward(torch.from_numpy(mel).to(dev), torch.tensor(lens).to(dev)).cpu()
    for b, n in enumerate(lens):
        ref = O.hifigan_forward(w, mel[b, :, :n], HIFIGAN_CONFIG)[0]
        assert maxabs(wave[b, :256 * n], ref) < WAVE_TOL, b
        assert float(wave[b, 256 * n:].abs().max()) == 0.0 if n < 23 else True


@pytest.mark.parametrize('mode,tol', [('f32', WAVE_TOL), ('bf16', 8e-3)])
def test_hifigan_ragged_fused_kernels_vs_oracle(dev, synth_weights, monkeypatch, mode, tol, ttsopt):
    """The fused c1 -> c2 pair and the all-phase transposed-conv kernels, FORCED ON, against the oracle on a ragged batch
    whose utterances end inside a fused tile's halo: fp32 resblock_pair tiles hold 252 / 248 / 244 outputs at C = 32 (252 at
    C = 64, k = 3) and the stage-4 / stage-3 lengths of 1- and 2-frame utterances (256, 512 / 128, 256) fall 4..12 columns
    past a tile edge; the bf16 octet engine's C = 128 tiles hold 254 / 250 / 246 outputs and a 4-frame utterance is 256
    positions there.  Every layer must zero-pad at the TRUE edge (models/fastpitch/networks.py:340-341 vocodes exact-length mels)."""
    import tts_oracle as O
    from ttsamd.config import HIFIGAN_CONFIG
    from ttsamd.engine import HifiGanEngine, set_precision
    ttsopt.set('TTSAMD_FUSED_PAIR', '1')
    ttsopt.set('TTSAMD_CONVT', '1')
    ttsopt.set('TTSAMD_BFO', '1')
    w = O.fold_weight_norm(synth_weights['hifigan'])
    rng = np.random.default_rng(11)
    lens = [19, 1, 2, 4, 8]
    mel = (rng.standard_normal((5, 80, 19)) * 1.5 - 4.0).astype(np.float32)
    set_precision(mode)
    try:
        wave = HifiGanEngine(synth_weights['hifigan'], device=dev).forward(torch.from_numpy(mel).to(dev), torch.tensor(lens).to(dev)).cpu()
    finally:
        set_precision('f32')
    for b, n in enumerate(lens):
        ref = O.hifigan_forward(w, mel[b, :, :n], HIFIGAN_CONFIG)[0]
        assert maxabs(wave[b, :256 * n], ref) < tol, (mode, b, maxabs(wave[b, :256 * n], ref))
        assert float(wave[b, 256 * n:].abs().max()) == 0.0 if n < 19 else True


@pytest.fixture(scope='module')
def fastpitch_engine(dev, synth_weights):
    from ttsamd.engine import FastPitchEngine
    return FastPitchEngine(synth_weights['fastpitch'])


def test_fastpitch_ragged_golden(dev, golden, fastpitch_engine):
    g = golden('fastpitch_b3_durtgt')
    mel, dec_lens, dur, pitch, energy = fastpitch_engine.infer(g['ids'], dur_tgt=g['dur_tgt'])
    assert np.array_equal(dec_lens.cpu().numpy(), g['dec_lens'])
    assert maxabs(dur, g['dur_pred']) < 1e-3
    assert maxabs(pitch, g['pitch_pred']) < 1e-3
    assert maxabs(energy, g['energy_pred']) < 1e-3
    assert maxabs(mel, g['mel']) < MEL_TOL


@pytest.mark.parametrize('tag', ['p1', 'p0p9_pitch'])
def test_fastpitch_predicted_durations_golden(dev, golden, fastpitch_engine, tag):
    g = golden(f'fastpitch_b2_pred_{tag}')
    out = fastpitch_engine.infer(g['ids'], pace=float(g['pace']), pitch_mul=float(g['pitch_mul']),
                                 pitch_add=float(g['pitch_add']), return_idx=True)
    mel, dec_lens, dur, pitch, energy, idx = out
    assert np.array_equal(dec_lens.cpu().numpy(), g['dec_lens'])       # bit-exact lengths
    assert maxabs(dur, g['dur_pred']) < 1e-3
    assert maxabs(pitch, g['pitch_pred']) < 1e-3
    assert maxabs(mel, g['mel']) < MEL_TOL


def test_fastpitch_multispeaker_golden(dev, golden, synth_weights):
    from ttsamd.engine import FastPitchEngine
    from ttsamd.config import NET_CONFIG
    g = golden('fastpitch_b3_spk2')
    eng = FastPitchEngine(synth_weights['fastpitch_spk4'], dict(NET_CONFIG, n_speakers=4))
    mel, dec_lens, *_ = eng.infer(g['ids'], dur_tgt=g['dur_tgt'], speaker=2)
    assert np.array_equal(dec_lens.cpu().numpy(), g['dec_lens'])
    assert maxabs(mel, g['mel']) < MEL_TOL


def test_end_to_end_golden_denoise_leg_torch_stft_standin(dev, golden, fastpitch_engine, hifigan_engine):
    """FastPitch2Wave.tts(list, batch_size=3, denoise=0) on three infer_text.txt lines."""
    e = golden('e2e_tts')
    t = golden('infer_text_ids')
    seqs = [t['flat'][t['offsets'][i]:t['offsets'][i + 1]] for i in e['line_idx']]
    order = np.argsort([-len(s) for s in seqs], kind='stable')
    ids = np.zeros((3, max(map(len, seqs))), np.int64)
    for r, i in enumerate(order):
        ids[r, :len(seqs[i])] = seqs[i]
    mel, dec_lens, *_ = fastpitch_engine.infer(ids)
    wave = hifigan_engine.forward(mel, dec_lens).cpu()
    dl = dec_lens.cpu().numpy()
    for r, i in enumerate(order):
        ref = e[f'wave{i}']
        assert 256 * dl[r] == ref.shape[0]
        assert maxabs(wave[r, :ref.shape[0]], ref) < WAVE_TOL


# ---------------------------------------------------------------------------------------
# The drop-in boundary: models.fastpitch.FastPitch2Wave / FastPitch, vocoder.load_hifigan,
# vocoder.hifigan.denoiser.Denoiser driven exactly as inference.py / test.py / app_utils.py do
# ---------------------------------------------------------------------------------------

@pytest.fixture(scope='module')
def checkpoints(tmp_path_factory, synth_weights):
    import json
    import text
    from ttsamd.config import NET_CONFIG, HIFIGAN_CONFIG
    d = tmp_path_factory.mktemp('ckpt')
    fp = {k: torch.from_numpy(v.copy()) for k, v in synth_weights['fastpitch'].items()}
    torch.save({'model': fp, 'config': dict(NET_CONFIG), 'symbols': list(text.symbols)}, d / 'fp.pth')
    torch.save({'generator': {k: torch.from_numpy(v.copy()) for k, v in synth_weights['hifigan'].items()}}, d / 'hg.pth')
    with open(d / 'config.json', 'w') as f:
        json.dump(HIFIGAN_CONFIG, f)
    return str(d / 'fp.pth'), str(d / 'hg.pth'), str(d / 'config.json')


def _lines(golden, idx):
    import json
    import os
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, 'infer_text_lines.json'), encoding='utf-8') as f:
        lines = json.load(f)
    return [lines[i] for i in idx]


def test_dropin_tts_matches_reference(dev, golden, checkpoints):
    from models.fastpitch import FastPitch2Wave
    e = golden('e2e_tts')
    texts = _lines(golden, e['line_idx'])
    model = FastPitch2Wave(checkpoints[0], vocoder_sd=checkpoints[1], vocoder_config=checkpoints[2])
    model = model.to(dev)
    model.eval()
    assert model.device.type == 'cuda'
    waves = model.tts(texts, batch_size=3, denoise=0.0)
    assert isinstance(waves, list) and all(w.device.type == 'cpu' and w.dim() == 1 for w in waves)
    for i, w in enumerate(waves):
        assert maxabs(w, e[f'wave{i}']) < WAVE_TOL
    # chunked path (batch_size < len) must equal the reference run on the same sub-batches: the
    # first chunk of 2 is what produced wave_dn* (denoise 0.005, tts default)
    wd = model.tts(texts[:2], batch_size=2)                      # default denoise=0.005
    for i, w in enumerate(wd):
        assert w.shape == e[f'wave_dn{i}'].shape
        assert maxabs(w, e[f'wave_dn{i}']) < WAVE_TOL
    # str input -> tts_single, return_mel
    w1, mel1 = model.tts(texts[0], denoise=0.0, return_mel=True)
    assert maxabs(w1, e['single_wave']) < WAVE_TOL
    assert maxabs(mel1, e['single_mel']) < MEL_TOL
    # batch_size == 1 loops tts_single
    w_list = model.tts(texts[:2], batch_size=1, denoise=0.0)
    assert maxabs(w_list[0], e['single_wave']) < WAVE_TOL
    # denoiser bias spectrum (torch.stft-based golden)
    assert maxabs(model.denoiser.bias_spec, e['bias_spec']) < 1e-4


def test_dropin_test_py_flow(dev, golden, checkpoints):
    """test.py:36-67: FastPitch(ckpt).ttmel(text) -> vocoder(mel[None]) -> denoiser(wave, s)."""
    import tts_oracle as O
    from models.fastpitch import FastPitch
    from vocoder import load_hifigan
    from vocoder.hifigan.denoiser import Denoiser
    e = golden('e2e_tts')
    texts = _lines(golden, e['line_idx'])
    model = FastPitch(checkpoints[0])
    vocoder = load_hifigan(state_dict_path=checkpoints[1], config_file=checkpoints[2])
    model, vocoder = model.to(dev), vocoder.to(dev)
    denoiser = Denoiser(vocoder)
    mel = model.ttmel(texts[0])
    assert maxabs(mel, e['single_mel']) < MEL_TOL
    wave = vocoder(mel[None])
    assert wave.shape == (1, 1, 256 * mel.shape[1])
    assert maxabs(wave[0, 0], e['single_wave']) < WAVE_TOL
    den = denoiser(wave, 0.01)
    ref = O.denoise(torch.from_numpy(e['single_wave'])[None], torch.from_numpy(e['bias_spec']), 0.01)
    assert maxabs(den[0], ref) < WAVE_TOL
    # ttmel(list, batch_size=2) returns mels in the original order
    mels = model.ttmel(texts, batch_size=3)
    assert [m.shape[1] * 256 for m in mels] == [e[f'wave{i}'].shape[0] for i in range(3)]


def test_dropin_cpu_device_raises(checkpoints):
    from models.fastpitch import FastPitch
    from ttsamd.lib import TtsAmdError
    model = FastPitch(checkpoints[0])           # stays on the CPU
    with pytest.raises(TtsAmdError):
        model.ttmel('marHabAF')


# ---------------------------------------------------------------------------------------
# MelVocos('22k') (config 5 back half)
# ---------------------------------------------------------------------------------------

def test_vocos_golden_and_ragged(dev, golden):
    import tts_oracle as O
    from ttsamd import synth
    from ttsamd.config import VOCOS_22K_CONFIG
    from vocoder.vocos import MelVocos
    g = golden('vocos_22k')
    w = synth.vocos_state_dict()
    voc = MelVocos('22k')
    voc.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    voc = voc.to(dev)
    assert maxabs(voc.bias_vec, g['bias_vec']) < 1e-4
    for T in (1, 5, 33):
        mel = torch.from_numpy(g[f'mel_T{T}']).to(dev)
        assert maxabs(voc(mel), g[f'wave_T{T}']) < WAVE_TOL
        assert maxabs(voc(mel, denoise=0.3), g[f'wave_dn_T{T}']) < WAVE_TOL
    # ragged batch == per-utterance exact-length runs of the oracle
    rng = np.random.default_rng(17)
    lens = [29, 7, 16]
    mel = (rng.standard_normal((3, 80, 29)) * 1.5 - 4.0).astype(np.float32)
    wave = voc(torch.from_numpy(mel).to(dev), lens=torch.tensor(lens).to(dev)).cpu()
    for b, n in enumerate(lens):
        ref = O.vocos_forward(w, mel[b:b + 1, :, :n], VOCOS_22K_CONFIG)[0]
        assert maxabs(wave[b, :256 * n], ref) < WAVE_TOL
        assert n == 29 or float(wave[b, 256 * n:].abs().max()) == 0.0


def test_fastpitch_multispeaker_plus_vocos(dev, golden, synth_weights):
    """Config 5 wiring (the build's own: the reference never connects them): 4-speaker FastPitch -> Vocos."""
    import tts_oracle as O
    from ttsamd import synth
    from ttsamd.config import NET_CONFIG, VOCOS_22K_CONFIG
    from ttsamd.engine import FastPitchEngine, VocosEngine
    g = golden('fastpitch_b3_spk2')
    fp = FastPitchEngine(synth_weights['fastpitch_spk4'], dict(NET_CONFIG, n_speakers=4))
    w = synth.vocos_state_dict()
    voc = VocosEngine(w)
    mel, dec_lens, *_ = fp.infer(g['ids'], dur_tgt=g['dur_tgt'], speaker=2)
    wave = voc.forward(mel, dec_lens).cpu()
    for b in range(3):
        n = int(g['dec_lens'][b])
        ref = O.vocos_forward(w, g['mel'][b:b + 1, :, :n], VOCOS_22K_CONFIG)[0]
        assert maxabs(wave[b, :256 * n], ref) < WAVE_TOL


def test_denoiser_strong_setting_torch_stft_standin_golden(dev, golden, hifigan_engine):
    """A large strength makes consecutive frames inconsistent, so every overlap-add term matters
    (a weak denoise is nearly the identity and hides frame-range bugs)."""
    import tts_oracle as O
    from ttsamd.engine import DenoiserEngine
    e = golden('e2e_tts')
    wave = torch.from_numpy(e['single_wave'])[None]
    bias = torch.from_numpy(e['bias_spec']) * 40.0
    ref = O.denoise(wave, bias, 1.0)
    eng = DenoiserEngine()
    out = eng.denoise(wave.to(dev).clone().contiguous(), torch.tensor([wave.shape[1]]).to(dev), bias, 1.0).cpu()
    assert float((ref - wave).abs().max()) > 1e-2          # the setting really changes the signal
    assert maxabs(out[:, :ref.shape[1]], ref) < WAVE_TOL


# ---------------------------------------------------------------------------------------
# bf16 MFMA modes (BASELINE config 3): stated tolerances, measured against the fp32 oracle
# ---------------------------------------------------------------------------------------
from conftest import BF16_MEL_TOL, BF16_WAVE_TOL   # noqa: E402  (plain bf16 operands: stated in tests/conftest.py)
X3_MEL_TOL, X3_WAVE_TOL = 1e-3, 1e-4            # split bf16 keeps the fp32 north-star tolerances


@pytest.fixture
def precision():
    from ttsamd.engine import set_precision
    yield set_precision
    set_precision('f32')


@pytest.mark.parametrize('mode,tol', [('bf16x3', 3e-5), ('bf16', 2e-2)])
def test_conv1d_kernel_bf16_modes(dev, precision, mode, tol):
    from ttsamd.engine import conv1d
    precision(mode)
    g = torch.Generator().manual_seed(5)
    for (cin, cout, k, dil, lin) in [(128, 128, 11, 5, 515), (32, 32, 3, 1, 300), (384, 1536, 3, 1, 64), (80, 512, 7, 1, 40)]:
        x = torch.randn(2, cin, lin, generator=g)
        w = torch.randn(cout, cin, k, generator=g) / np.sqrt(cin * k)
        b = torch.randn(cout, generator=g)
        y = conv1d(x.to(dev), w.to(dev), b.to(dev), dilation=dil, in_slope=0.1).cpu()
        ref = torch.nn.functional.conv1d(torch.nn.functional.leaky_relu(x.double(), 0.1), w.double(), b.double(),
                                         dilation=dil, padding=(k * dil - dil) // 2)
        assert maxabs(y, ref) < tol, (mode, cin, cout, k)


@pytest.mark.parametrize('mode,tol', [('bf16x3', 5e-5), ('bf16', 4e-2)])
def test_conv1d_kernel_bf16_modes_large_tiles(dev, precision, mode, tol):
    precision(mode)
    _conv_fuzz(dev, _large_tile_cases(7, ks=(3, 11)), tol, 7)


@pytest.mark.parametrize('mode,mel_tol,wave_tol', [('bf16x3', X3_MEL_TOL, X3_WAVE_TOL), ('bf16', BF16_MEL_TOL, BF16_WAVE_TOL)])
def test_end_to_end_bf16_modes(dev, golden, precision, fastpitch_engine, hifigan_engine, mode, mel_tol, wave_tol):
    precision(mode)
    g = golden('fastpitch_b3_durtgt')
    mel, dec_lens, *_ = fastpitch_engine.infer(g['ids'], dur_tgt=g['dur_tgt'])
    assert np.array_equal(dec_lens.cpu().numpy(), g['dec_lens'])
    err_mel = maxabs(mel, g['mel'])
    h = golden('hifigan_T40')
    wave = hifigan_engine.forward(torch.from_numpy(h['mel'])[None].to(dev))
    err_wave = maxabs(wave[0], h['wave'][0])
    print(f'{mode}: mel max-abs {err_mel:.2e}, wave max-abs {err_wave:.2e}')
    assert err_mel < mel_tol and err_wave < wave_tol


# ---------------------------------------------------------------------------------------
# Edge cases and C-ABI error behaviour
# ---------------------------------------------------------------------------------------

def test_edge_cases_lengths(dev, synth_weights, fastpitch_engine, hifigan_engine):
    import tts_oracle as O
    from ttsamd.config import NET_CONFIG, HIFIGAN_CONFIG
    fw = O.to_torch(synth_weights['fastpitch'])
    # single token, single utterance
    ids = np.array([[7]], np.int64)
    mel, dec_lens, dur, *_ = fastpitch_engine.infer(ids)
    rmel, rlens, rdur, *_ = O.fastpitch_infer(fw, NET_CONFIG, ids)
    assert np.array_equal(dec_lens.cpu().numpy(), rlens.numpy()) and maxabs(mel, rmel) < MEL_TOL
    # max_duration clamp and a zero-duration token inside the utterance (reps = 0 -> token skipped)
    ids = np.array([[3, 9, 4, 12, 0, 0], [5, 6, 0, 0, 0, 0]], np.int64)
    dur_tgt = np.array([[80.0, 0.0, 2.0, 0.4, 0, 0], [1.0, 0.49, 0, 0, 0, 0]], np.float32)
    mel, dec_lens, *_ , idx = fastpitch_engine.infer(ids, dur_tgt=dur_tgt, return_idx=True)
    rmel, rlens, *_ = O.fastpitch_infer(fw, NET_CONFIG, ids, dur_tgt=dur_tgt)
    _, _, ridx = O.regulate_len_indices(dur_tgt, 1.0)
    assert np.array_equal(dec_lens.cpu().numpy(), rlens.numpy()) and np.array_equal(idx.cpu().numpy(), ridx)
    assert maxabs(mel, rmel) < MEL_TOL
    # vocoder: an utterance of length 0 inside a batch produces silence and does not disturb the others
    hw = O.fold_weight_norm(synth_weights['hifigan'])
    rng = np.random.default_rng(23)
    melv = (rng.standard_normal((3, 80, 6)) * 1.5 - 4.0).astype(np.float32)
    lens = torch.tensor([6, 0, 2])
    wave = hifigan_engine.forward(torch.from_numpy(melv).to(dev), lens.to(dev)).cpu()
    assert float(wave[1].abs().max()) == 0.0
    for b in (0, 2):
        n = int(lens[b])
        assert maxabs(wave[b, :256 * n], O.hifigan_forward(hw, melv[b, :, :n], HIFIGAN_CONFIG)[0]) < WAVE_TOL


def test_longest_infer_text_line(dev, golden, synth_weights, fastpitch_engine, hifigan_engine):
    """Config-1 extreme: the longest data/infer_text.txt line (268 tokens, ~2000 frames) vs the oracle."""
    import tts_oracle as O
    from ttsamd.config import NET_CONFIG, HIFIGAN_CONFIG
    t = golden('infer_text_ids')
    lens = np.diff(t['offsets'])
    i = int(np.argmax(lens))
    ids = t['flat'][t['offsets'][i]:t['offsets'][i + 1]][None]
    assert ids.shape[1] == 268
    dur = (1 + (np.arange(268) % 9)).astype(np.float32)[None]          # forced durations: index parity exact
    mel, dec_lens, *_ = fastpitch_engine.infer(ids, dur_tgt=dur)
    rmel, rlens, *_ = O.fastpitch_infer(O.to_torch(synth_weights['fastpitch']), NET_CONFIG, ids, dur_tgt=dur)
    assert np.array_equal(dec_lens.cpu().numpy(), rlens.numpy())
    assert maxabs(mel, rmel) < MEL_TOL
    wave = hifigan_engine.forward(mel, dec_lens).cpu()
    ref = O.hifigan_forward(O.fold_weight_norm(synth_weights['hifigan']), rmel[0], HIFIGAN_CONFIG)[0]
    assert maxabs(wave[0], ref) < WAVE_TOL


def test_c_abi_error_codes(dev, hifigan_engine):
    """Negative return codes + thread-local message instead of crashes."""
    import ctypes as C
    from ttsamd import lib as L
    lib = L.load()
    mel = torch.zeros(1, 80, 4, device=dev)
    wave = torch.zeros(1, 1024, device=dev)
    ws = torch.empty(16, dtype=torch.uint8, device=dev)            # far too small
    rc = lib.ttsamd_hifigan_forward(hifigan_engine.handle, C.c_void_p(mel.data_ptr()), C.c_void_p(0), 1, 4,
                                    C.c_void_p(wave.data_ptr()), C.c_void_p(ws.data_ptr()), 16, C.c_void_p(0))
    assert rc == -3 and b'workspace' in lib.ttsamd_last_error()
    rc = lib.ttsamd_hifigan_forward(hifigan_engine.handle, C.c_void_p(0), C.c_void_p(0), 1, 4, C.c_void_p(0),
                                    C.c_void_p(0), 0, C.c_void_p(0))
    assert rc == -1
    assert lib.ttsamd_set_precision(7) == -1
    h = C.c_void_p()
    cfg = L.HifiGanCfg()
    cfg.n_ups, cfg.n_kernels, cfg.n_dilations = 4, 3, 3
    arr, keep = L.make_tensors({})
    assert lib.ttsamd_hifigan_create(arr, 0, C.byref(cfg), C.byref(h)) == -1      # missing tensors
    assert b'conv_pre' in lib.ttsamd_last_error()


def test_full_size_bench_workload_properties(dev, synth_weights, fastpitch_engine, hifigan_engine):
    """BASELINE config 2 at its full size (32 utterances x 64 tokens, the bench.py workload; single-stream
    vocoder, large tiles).  The oracle needs ~1 s per utterance, so it checks two utterances; the rest is
    covered by size-independent properties: run-to-run determinism, frame counts = sum of the durations,
    silence past each utterance's end, |wave| <= 1, and batch independence (the same utterance synthesised
    alone goes through the small-batch path: split-K, three streams, other tiles)."""
    import tts_oracle as O
    from ttsamd import synth
    from ttsamd.config import NET_CONFIG, HIFIGAN_CONFIG
    B, Lt = 32, 64
    ids = torch.from_numpy(synth.synth_ids(B, Lt)).to(dev)
    dur = torch.from_numpy(synth.synth_durations(B, Lt)).to(dev)
    mel, dec_lens, *_ = fastpitch_engine.infer(ids, dur_tgt=dur)
    wave = hifigan_engine.forward(mel, dec_lens)
    mel2, dec_lens2, *_ = fastpitch_engine.infer(ids, dur_tgt=dur)
    wave2 = hifigan_engine.forward(mel2, dec_lens2)
    assert torch.equal(mel, mel2) and torch.equal(wave, wave2)                  # deterministic
    dl = dec_lens.cpu().numpy()
    assert np.array_equal(dl, np.floor(dur.cpu().numpy() + 0.5).astype(np.int64).sum(1))
    assert wave.shape == (B, 256 * int(dl.max())) and bool(torch.isfinite(wave).all())
    assert float(wave.abs().max()) <= 1.0
    w = wave.cpu()
    for b in range(B):
        assert float(w[b, 256 * int(dl[b]):].abs().max()) == 0.0 if dl[b] < dl.max() else True
    # batch independence of the vocoder: an utterance's mel vocoded alone (small-batch path) gives the same wave
    for b in (0, 13, 31):
        n = int(dl[b])
        w1 = hifigan_engine.forward(mel[b:b + 1, :, :n].contiguous(), dec_lens[b:b + 1])
        assert maxabs(w1[0], wave[b, :256 * n]) < WAVE_TOL
    # FastPitch follows the reference's padded-batch semantics (the conv-FF hidden layer is not
    # masked, so an utterance's last frames see relu(bias) from the first pad frame when a longer utterance
    # shares the batch, and zero padding when it is the longest).  Only the longest utterance is therefore
    # batch-independent; the oracle sub-batch below contains it for the same reason.
    bmax = int(np.argmax(dl))
    m1, l1, *_ = fastpitch_engine.infer(ids[bmax:bmax + 1], dur_tgt=dur[bmax:bmax + 1])
    assert int(l1[0]) == int(dl[bmax]) and maxabs(m1[0], mel[bmax]) < 1e-4
    fw, hw = O.to_torch(synth_weights['fastpitch']), O.fold_weight_norm(synth_weights['hifigan'])
    sel = [5 if bmax != 5 else 6, bmax]
    assert dl[sel[0]] < dl[bmax]
    with torch.inference_mode():
        mel_ref, lens_ref, waves_ref = O.tts_batch(fw, NET_CONFIG, hw, HIFIGAN_CONFIG, ids[sel].cpu().numpy(),
                                                   dur_tgt=dur[sel].cpu().numpy())
    for r, b in enumerate(sel):
        n = int(dl[b])
        assert int(lens_ref[r]) == n
        assert maxabs(mel[b, :, :n], np.asarray(mel_ref[r])[:, :n]) < MEL_TOL
        assert maxabs(w[b, :256 * n], waves_ref[r]) < WAVE_TOL


def test_hifigan_branch_streams_bit_identical(dev, hifigan_engine, monkeypatch, ttsopt):
    """Small batches run the three ResBlocks of a stage on three streams (csrc/hifigan.hip); the accumulation
    into the stage output is event-chained in the reference's order, so the waves must equal the one-stream
    result bit for bit, call after call (a missing dependency would show up as run-to-run differences)."""
    rng = np.random.default_rng(11)
    lens = torch.tensor([37, 12, 30]).to(dev)
    mel = torch.from_numpy((rng.standard_normal((3, 80, 37)) * 1.5 - 4.0).astype(np.float32)).to(dev)
    ttsopt.set('TTSAMD_HIFIGAN_STREAMS', '0')
    ref = hifigan_engine.forward(mel, lens).clone()
    ttsopt.set('TTSAMD_HIFIGAN_STREAMS', '1')
    for _ in range(20):
        assert torch.equal(hifigan_engine.forward(mel, lens), ref)
    ttsopt.set('TTSAMD_HIFIGAN_STREAMS', None)
    assert torch.equal(hifigan_engine.forward(mel, lens), ref)


def test_dropin_app_utils_request_flow(dev, golden, checkpoints):
    """utils/app_utils.py:62-81 (TTSManager.tts): per request model.to(device) -> ttmel -> vocoder ->
    denoiser -> peak-normalise to 0.99 -> model.cpu().  The second request must reuse the device handle
    (no weight re-upload) and give the same audio."""
    from models.fastpitch import FastPitch
    from vocoder import load_hifigan
    from vocoder.hifigan.denoiser import Denoiser
    from utils.audio import peak_normalise
    e = golden('e2e_tts')
    text = _lines(golden, e['line_idx'])[0]
    model = FastPitch(checkpoints[0])
    vocoder = load_hifigan(checkpoints[1], checkpoints[2])
    denoiser = Denoiser(vocoder, mode='zeros')
    vocoder.to(dev)
    denoiser.to(dev)
    outs, engines = [], []
    for _ in range(2):
        model.to(dev)
        mel = model.ttmel(text, speed=1)
        wave = vocoder(mel)
        wave_den = denoiser(wave, 0.01)
        wave_den = wave_den / wave_den.abs().max() * 0.99
        outs.append(wave_den.cpu())
        engines.append(model.engine())
        model.cpu()
    assert engines[0] is engines[1]
    assert torch.equal(outs[0], outs[1])
    assert abs(float(outs[0].abs().max()) - 0.99) < 1e-6
    assert maxabs(peak_normalise(wave_den.cpu().reshape(-1)), outs[0].reshape(-1)) < 1e-6


def test_bf16_packed_intermediate_bit_identical(dev, precision, hifigan_engine, monkeypatch, ttsopt):
    """bf16 mode: the c1 -> c2 intermediate of every ResBlock crosses HBM as packed bf16 (leaky-relu and RNE
    rounding done by the producer instead of the consumer's staging): same rounding point, same bits.
    Covers the small tiles (B=3) and the large ones (B=24 x 200 frames)."""
    precision('bf16')
    rng = np.random.default_rng(5)
    for B, T in ((3, 37), (24, 200)):
        lens = torch.from_numpy(rng.integers(T // 2, T + 1, size=B)).to(dev)
        lens[0] = T
        mel = torch.from_numpy((rng.standard_normal((B, 80, T)) * 1.5 - 4.0).astype(np.float32)).to(dev)
        ttsopt.set('TTSAMD_BF16_PACKED_T', '0')
        ref = hifigan_engine.forward(mel, lens).clone()
        ttsopt.set('TTSAMD_BF16_PACKED_T', '1')
        out = hifigan_engine.forward(mel, lens)
        assert torch.equal(out, ref), (B, T, float((out - ref).abs().max()))


# ---------------------------------------------------------------------------------------
# error behaviour at the edges the reference raises on: indices, tiny utterances
# ---------------------------------------------------------------------------------------
def test_out_of_range_ids_raise_like_nn_embedding(dev, checkpoints):
    from models.fastpitch import FastPitch
    model = FastPitch(checkpoints[0]).to(dev)
    n = model.net_config['n_symbols']
    with pytest.raises(IndexError):
        model.infer(torch.tensor([[1, 2, n]]))
    with pytest.raises(IndexError):
        model.infer(torch.tensor([[1, -1, 2]]))
    mel, dec_lens, *_ = model.infer(torch.tensor([[1, 2, n - 1]]))          # the last valid id is fine
    assert mel.shape[0] == 1 and int(dec_lens[0]) == mel.shape[2]


def test_negative_forced_durations_give_zero_repeats(dev, fastpitch_engine):
    ids = torch.tensor([[3, 4, 5, 6]])
    dur = torch.tensor([[2.0, -3.0, 1.0, -0.2]])
    mel, dec_lens, *_ = fastpitch_engine.infer(ids, dur_tgt=dur)
    assert int(dec_lens[0]) == 3 and mel.shape[2] == 3 and bool(torch.isfinite(mel).all())


def test_denoiser_ragged_batch_equals_the_oracle_per_utterance(dev):
    """`denoise_fft_kernel` (one block per frame: window x reflect-padded samples -> FFT-1024 in LDS -> spectral subtraction -> inverse FFT) +
    the overlap-add on a ragged batch -- lengths that are and are not multiples of the hop, one just above the reflect-pad minimum, strong
    setting: every row equals the oracle's torch.stft / istft run on that row alone (vocoder/hifigan/denoiser.py:66-72); samples past a
    row's length stay untouched."""
    import tts_oracle as O
    from ttsamd.engine import DenoiserEngine
    g = torch.Generator().manual_seed(11)
    ns = [256 * 37, 256 * 12 + 100, 513, 256 * 40, 3000]
    wave = torch.randn(len(ns), max(ns), generator=g) * 0.2
    bias = torch.rand(1, 513, 1, generator=g) * 3.0
    eng = DenoiserEngine(device=dev)
    out = eng.denoise(wave.to(dev).clone().contiguous(), torch.tensor(ns).to(dev), bias.to(dev), 1.0).cpu()
    worst = 0.0
    for b, n in enumerate(ns):
        ref = O.denoise(wave[b:b + 1, :n], bias, 1.0)
        m = ref.shape[1]                                                   # istft(center) returns hop * (frames - 1) samples
        worst = max(worst, float((out[b, :m] - ref[0]).abs().max()))
        assert float((ref[0] - wave[b, :m]).abs().max()) > 1e-2           # the setting changes the signal
        assert torch.equal(out[b, n:], wave[b, n:])
    print(f'ragged denoise, 5 rows: max-abs {worst:.2e} (tol {WAVE_TOL})')
    assert worst < WAVE_TOL


def test_denoiser_rejects_utterances_of_at_most_512_samples(dev, hifigan_engine):
    """torch's reflect pad (Spectrogram(center=True), denoiser.py:43-48) raises when n <= n_fft/2; the ragged batch
    path must not read outside the row instead (2 mel frames = 512 samples next to a normal utterance)."""
    from vocoder.hifigan.denoiser import Denoiser
    from vocoder.hifigan.models import Generator
    from ttsamd.engine import DenoiserEngine
    eng = DenoiserEngine(device=dev)
    wave = torch.randn(2, 4096, device=dev) * 0.1
    ns = torch.tensor([4096, 512], device=dev)
    bias = torch.rand(513, device=dev) * 0.01
    ref = wave.clone()
    out = eng.denoise(wave.clone(), ns, bias, 0.1)                          # engine level: memory-safe, finite
    assert bool(torch.isfinite(out).all()) and torch.equal(out[1, 512:], ref[1, 512:])
    solo = eng.denoise(ref[:1].clone(), ns[:1], bias, 0.1)
    assert torch.equal(out[0], solo[0])                                     # the short neighbour does not leak into row 0

    class _Voc:                                                             # wrapper level: raises like the reference
        device = dev

        def __call__(self, mel):
            return hifigan_engine.forward(mel)
    den = Denoiser.__new__(Denoiser)
    torch.nn.Module.__init__(den)
    with pytest.raises(ValueError):
        Denoiser.forward_batch(den, wave.clone(), ns, 0.1)


def test_fused_pair_and_all_phase_convt_match_the_generic_engine(dev, synth_weights, hifigan_engine, monkeypatch, ttsopt):
    """Round-2 kernels against the generic MFMA conv engine they replace, same weights, ragged batch with a 1-frame and a
    2-tile utterance: `resblock_pair<K, C>` (c1 -> c2 of the C = 32 stage, and of the C = 64 stage at k = 3, in one launch, intermediate in LDS, halo recompute)
    and `convt_mfma_f32` (all output phases of a transposed conv per wave).  Only the summation order differs (bias first,
    residual in the accumulator), so the waves agree far inside the waveform tolerance; each schedule is bit-reproducible."""
    rng = np.random.default_rng(21)
    lens = torch.tensor([41, 1, 17, 2]).to(dev)
    mel = torch.from_numpy((rng.standard_normal((4, 80, 41)) * 1.5 - 4.0).astype(np.float32)).to(dev)
    ttsopt.set('TTSAMD_FUSED_PAIR', '0')
    ttsopt.set('TTSAMD_CONVT', '0')
    ref = hifigan_engine.forward(mel, lens).clone()
    for fused, convt in (('1', '0'), ('0', '1'), ('1', '1')):
        ttsopt.set('TTSAMD_FUSED_PAIR', fused)
        ttsopt.set('TTSAMD_CONVT', convt)
        out = hifigan_engine.forward(mel, lens).clone()
        assert maxabs(out, ref) < 5e-6, (fused, convt)
        assert torch.equal(hifigan_engine.forward(mel, lens), out)
        for b in range(4):
            n = 256 * int(lens[b])
            assert float(out[b, n:].abs().max()) == 0.0 if n < out.shape[1] else True
    ttsopt.set('TTSAMD_FUSED_PAIR', None)
    ttsopt.set('TTSAMD_CONVT', None)
    assert torch.equal(hifigan_engine.forward(mel, lens), out)          # both are the default


def test_two_stream_pipeline_bit_identical(dev, synth_weights, fastpitch_engine, hifigan_engine):
    """ttsamd.pipeline.FastPitchHifiGan: FastPitch of batch i + 1 on its own stream under HiFi-GAN of batch i gives the same
    waves, bit for bit, as issuing the two stages of each batch on one stream."""
    from ttsamd import synth
    from ttsamd.pipeline import FastPitchHifiGan
    ids = torch.from_numpy(synth.synth_ids(6, 20)).to(dev)
    dur = torch.from_numpy(synth.synth_durations(6, 20)).to(dev)
    ref = []
    for i in range(3):
        mel, dl, *_ = fastpitch_engine.infer(ids[2 * i:2 * i + 2], dur_tgt=dur[2 * i:2 * i + 2])
        ref.append((hifigan_engine.forward(mel, dl).clone(), dl.clone()))
    pipe = FastPitchHifiGan(fastpitch_engine, hifigan_engine, dev)
    got = [pipe.submit(ids[2 * i:2 * i + 2], dur_tgt=dur[2 * i:2 * i + 2]) for i in range(3)]
    pipe.join()
    torch.cuda.synchronize()
    for (w_ref, dl_ref), (_, dl, w) in zip(ref, got):
        assert torch.equal(dl, dl_ref) and torch.equal(w, w_ref)


def test_dropin_tts_list_pipeline_matches_the_one_stream_loop(dev, golden, checkpoints, monkeypatch, capsys):
    """`FastPitch2Wave.tts(list)` over several chunks runs as a three-stream pipeline (FastPitch of the next chunks under the vocoder
    of the previous ones, D2H on a third stream) whose vocoder takes the mels of up to 16 utterances per ragged call: same lengths
    and -- the vocoder being batch-independent up to its fp32 summation order -- the same waves as the one-stream loop to 1e-5
    (north-star tolerance 1e-4), for batch_size 1 (tts_single per line), 2 and 5 (tts_batch), denoiser on; prints both timings.
    batch_size 1 additionally runs FastPitch itself on ragged groups of length-sorted lines whose rows are computed as if alone (engine batch
    mode 1, tests/test_gpu_alone.py; TTSAMD_TTS_ALONE=0 = the line-by-line FastPitch calls under the same pipeline): all three agree.
    (Both paths are checked against the oracle: test_dropin_tts_matches_reference, test_config1_all_100_lines_batch_size_1.)"""
    import time
    from models.fastpitch import FastPitch2Wave
    lines = _lines(golden, range(24))
    model = FastPitch2Wave(checkpoints[0], vocoder_sd=checkpoints[1], vocoder_config=checkpoints[2]).to(dev)
    for bs in (1, 2, 5):
        res = {}
        for mode in ('0', '1'):
            monkeypatch.setenv('TTSAMD_TTS_PIPELINE', mode)
            model.tts(lines[:4], batch_size=bs)                              # warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res[mode] = (model.tts(lines, batch_size=bs), time.perf_counter() - t0)
        a, b = res['0'][0], res['1'][0]
        assert len(a) == len(b) == len(lines)
        for x, y in zip(a, b):
            assert x.device.type == 'cpu' and x.shape == y.shape and float((x - y).abs().max()) < 1e-5
        if bs == 1:
            monkeypatch.setenv('TTSAMD_TTS_ALONE', '0')
            c = model.tts(lines, batch_size=bs)
            monkeypatch.delenv('TTSAMD_TTS_ALONE')
            for x, y in zip(a, c):
                assert x.shape == y.shape and float((x - y).abs().max()) < 1e-5
        with capsys.disabled():
            print(f'\n[tts list, {len(lines)} lines, batch_size {bs}] one stream {res["0"][1] * 1e3:.1f} ms, pipelined {res["1"][1] * 1e3:.1f} ms')


def test_fastpitch_deep_splitk_tiles(dev, fastpitch_engine, monkeypatch, ttsopt):
    """Batch 8 x 64 tokens (~450 frames each): the second conv-FF conv of the decoder (1536 -> 384) runs as 128 x 64 tiles with K
    split into slices + a reduce launch; same mel as the un-split 64 x 64 tiles up to the summation order."""
    from ttsamd import synth
    ids = torch.from_numpy(synth.synth_ids(8, 64)).to(dev)
    dur = torch.from_numpy(synth.synth_durations(8, 64)).to(dev)
    ttsopt.set('TTSAMD_WINO4', '14')            # k = 3 off the F(4,3) kernel (which would take this shape with its own split K): the direct kernel's tiles
    ttsopt.set('TTSAMD_DEEP_SPLITK', '0')
    mel0, dl0, *_ = fastpitch_engine.infer(ids, dur_tgt=dur)
    ttsopt.set('TTSAMD_DEEP_SPLITK', '1')
    mel1, dl1, *_ = fastpitch_engine.infer(ids, dur_tgt=dur)
    assert torch.equal(dl0, dl1) and bool(torch.isfinite(mel1).all())
    assert maxabs(mel0, mel1) < 2e-5
    assert not torch.equal(mel0, mel1)          # the other schedule did run


def test_attention_schedules_bit_identical(dev, fastpitch_engine, monkeypatch, ttsopt):
    """FastPitch's self-attention (transformer.py:131-141) merges one tile-local softmax per 64-key tile in tile order; which block
    does it is a schedule: 64 or 16 queries per block walking the tiles (TTSAMD_ATT_RA), or one block per (16 queries, key tile)
    plus a merge launch (the batch-1 default).  Same mel bits on a ragged batch whose decoder sequences span 1 ... 8 key tiles."""
    from ttsamd import synth
    ids = synth.synth_ids(3, 64)
    ids[1, 9:] = 0
    ids[2, 30:] = 0
    dur = synth.synth_durations(3, 64) * (ids != 0)
    ids_d, dur_d = torch.from_numpy(ids).to(dev), torch.from_numpy(dur).to(dev)
    mels = []
    for ra, split in (('4', '0'), ('1', '0'), ('2', '0'), ('1', '1')):
        ttsopt.set('TTSAMD_ATT_RA', ra)
        ttsopt.set('TTSAMD_ATT_SPLIT', split)
        mel, lens, *_ = fastpitch_engine.infer(ids_d, dur_tgt=dur_d)
        mels.append(mel.cpu())
    assert int(lens.max()) > 128 and int(lens.min()) < 64 and bool(torch.isfinite(mels[0]).all())
    for m in mels[1:]:
        assert torch.equal(m, mels[0])


@pytest.mark.parametrize('cin,cout,k,dil,lin,B', [(256, 256, 7, 3, 3584, 16), (128, 128, 11, 5, 7168, 16), (384, 1536, 3, 1, 448, 32), (1536, 384, 3, 1, 512, 16)])
def test_conv1d_block_order_maps_bit_identical(dev, monkeypatch, cin, cout, k, dil, lin, B, ttsopt):
    """Which XCD runs which (time tile, co-tile) is a schedule: the tile-owning map (an XCD keeps the co-tiles of its time tiles: default), one
    co-tile class per XCD (TTSAMD_XCD_WMAX_KB=0), every co-tile on one XCD and the plain grid order (TTSAMD_XCD_W=0) give the same bits on a
    ragged batch with an empty utterance, and the float64 result within the conv tolerance."""
    from ttsamd.engine import conv1d
    g = torch.Generator().manual_seed(cin + cout + k + lin)
    x = torch.randn(B, cin, lin, generator=g)
    w = torch.randn(cout, cin, k, generator=g) / np.sqrt(cin * k)
    b = torch.randn(cout, generator=g)
    rng = np.random.default_rng(lin)
    lens = torch.from_numpy(rng.integers(lin // 2, lin + 1, size=B)).long()
    lens[0], lens[1], lens[B - 1] = lin, 0, lin - 3
    xd, wd, bd, ld = x.to(dev), w.to(dev), b.to(dev), lens.to(dev)
    ys = []
    for wmax, xw in (('3000', '1'), ('0', '1'), ('100000', '1'), ('3000', '0')):
        ttsopt.set('TTSAMD_XCD_WMAX_KB', wmax)
        ttsopt.set('TTSAMD_XCD_W', xw)
        ys.append(conv1d(xd, wd, bd, ld, dilation=dil, in_slope=0.1).cpu())
    assert all(torch.equal(ys[0], y) for y in ys[1:])
    for i in (0, 1, 2, B - 1):
        n = int(lens[i])
        if n:
            ref = torch.nn.functional.conv1d(torch.nn.functional.leaky_relu(x[i:i + 1, :, :n].double(), 0.1), w.double(), b.double(),
                                             dilation=dil, padding=(k * dil - dil) // 2)[0]
            assert maxabs(ys[0][i, :, :n], ref) < 3e-5, (i, n)
        assert n == lin or float(ys[0][i, :, n:].abs().max()) == 0.0
