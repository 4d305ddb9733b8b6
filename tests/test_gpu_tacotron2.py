"""GPU checks of the Tacotron2 path (BASELINE config 4) through the C ABI against
oracle/taco_oracle.py on the same seeded weights and inputs.

PARITY UNPINNED: the oracle restates torchaudio.models.tacotron2 (not vendored by the reference,
not installed here) from its published architecture, so these tests prove HIP == restatement, not
HIP == reference.  Tolerances: mel 1e-3 max-abs (north_star), alignments 1e-4, lengths exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

MEL_TOL = 1e-3
ALIGN_TOL = 1e-4


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


def _weights(gate_bias, num_speakers=40, seed=0):
    from ttsamd.config import TACOTRON2_CONFIG
    from ttsamd.synth import tacotron2_state_dict
    cfg = dict(TACOTRON2_CONFIG, num_speakers=num_speakers)
    return cfg, tacotron2_state_dict(cfg, seed=seed, gate_bias=gate_bias)


def _tokens(B, L, seed):
    g = torch.Generator().manual_seed(seed)
    lens = torch.sort(torch.randint(max(1, L // 2), L + 1, (B,), generator=g), descending=True).values
    lens[0] = L
    tok = torch.randint(1, 40, (B, L), generator=g)
    tok = tok * (torch.arange(L)[None] < lens[:, None])
    return tok, lens


@pytest.mark.parametrize('B,L,steps,seed', [(3, 23, 20, -1), (2, 40, 12, 11), (1, 7, 9, 0)])
def test_tacotron2_matches_oracle(dev, B, L, steps, seed):
    """gate never fires (bias -20): the loop runs to max_step; mel, lengths and alignments vs the oracle,
    with the prenet dropout off (seed -1) and on (hash masks shared with the oracle)."""
    import taco_oracle as T
    from ttsamd.engine import Tacotron2Engine
    cfg, sd = _weights(gate_bias=-20.0)
    tok, lens = _tokens(B, L, 100 + B)
    sids = torch.arange(B) % cfg['num_speakers']
    mel_ref, lens_ref, al_ref = T.tacotron2_infer(sd, cfg, tok, sids, lens, max_step=steps, seed=seed)
    eng = Tacotron2Engine(sd, cfg, device=dev)
    mel, mel_lens, al = eng.infer(tok, sids, lens, max_step=steps, dropout_seed=seed)
    assert mel.shape == (B, 80, steps) and al.shape == (B, steps, L)
    assert mel_lens.cpu().tolist() == np.asarray(lens_ref).tolist() == [steps] * B
    assert maxabs(al, al_ref) < ALIGN_TOL
    assert maxabs(mel, mel_ref) < MEL_TOL
    # attention rows are distributions over the valid tokens only
    a = al.cpu()
    assert torch.allclose(a.sum(-1), torch.ones(B, steps), atol=1e-5)
    for b in range(B):
        assert float(a[b, :, int(lens[b]):].abs().max()) == 0.0 if int(lens[b]) < L else True


def test_tacotron2_single_speaker_model(dev):
    """num_speakers=1: no speaker embedding, memory dim 512 (tacotron2_ms.py:190-193,311-318)."""
    import taco_oracle as T
    from ttsamd.engine import Tacotron2Engine
    cfg, sd = _weights(gate_bias=-20.0, num_speakers=1, seed=3)
    assert 'speaker_embedding.weight' not in sd
    tok, lens = _tokens(2, 17, 5)
    mel_ref, _, al_ref = T.tacotron2_infer(sd, cfg, tok, None, lens, max_step=10, seed=-1)
    mel, _, al = Tacotron2Engine(sd, cfg, device=dev).infer(tok, None, lens, max_step=10, dropout_seed=-1)
    assert maxabs(mel, mel_ref) < MEL_TOL and maxabs(al, al_ref) < ALIGN_TOL


def _gate_for_stops(cfg, sd, tok, sids, lens, stops, max_step, seed):
    """Synthetic gates barely move in time, so fit (ridge, dual form) a gate layer whose logit is -2 before
    utterance b's chosen stop step and +2 from it on.  The trajectory before a stop does not depend on the
    gate, so a traced never-stopping run provides the layer's inputs; with the prenet dropout on they differ
    enough from step to step that the fit needs only O(1) weights (margin 2 vs ~1e-3 of fp32 noise)."""
    import taco_oracle as T
    tr = {}
    T.tacotron2_infer(sd, cfg, tok, sids, lens, max_step=max_step, seed=seed, trace=tr)
    hc = torch.stack(tr['hc']).double()                                   # [T, B, D+M]
    rows, tgt = [], []
    for b, stop in enumerate(stops):
        for t in range(max_step):
            rows.append(hc[t, b])
            tgt.append(2.0 if t >= stop - 1 else -2.0)
    H, y = torch.stack(rows), torch.tensor(tgt, dtype=torch.float64)
    w = H.T @ torch.linalg.solve(H @ H.T + 1e-6 * torch.eye(len(y), dtype=torch.float64), y)
    assert float(((H @ w) * y).min()) > 1.0 and float(w.abs().sum()) < 5e3
    out = dict(sd)
    out['decoder.gate_layer.weight'] = w[None].float().numpy().copy()
    out['decoder.gate_layer.bias'] = np.zeros(1, np.float32)
    return out


def test_tacotron2_stop_token(dev):
    """Utterances stop at different steps: per-utterance lengths, the early exit (T = max length, not a
    multiple of the engine's 8-step polling) and the frames computed after an utterance finished all match."""
    import taco_oracle as T
    from ttsamd.engine import Tacotron2Engine
    cfg, sd = _weights(gate_bias=-20.0)
    tok, lens = _tokens(4, 19, 9)
    sids = torch.tensor([0, 3, 7, 39])
    stops = [12, 27, 5, 21]
    sd = _gate_for_stops(cfg, sd, tok, sids, lens, stops, max_step=40, seed=4)
    mel_ref, lens_ref, al_ref = T.tacotron2_infer(sd, cfg, tok, sids, lens, max_step=40, seed=4)
    assert np.asarray(lens_ref).tolist() == stops and mel_ref.shape[2] == 27
    mel, mel_lens, al = Tacotron2Engine(sd, cfg, device=dev).infer(tok, sids, lens, max_step=40, dropout_seed=4)
    assert mel_lens.cpu().tolist() == stops
    assert mel.shape == tuple(mel_ref.shape)
    assert maxabs(mel, mel_ref) < MEL_TOL and maxabs(al, al_ref) < ALIGN_TOL


def test_tacotron2_max_step_and_errors(dev):
    from ttsamd.engine import Tacotron2Engine
    from ttsamd.lib import TtsAmdError
    cfg, sd = _weights(gate_bias=-20.0)
    eng = Tacotron2Engine(sd, cfg, device=dev)
    tok, lens = _tokens(2, 9, 1)
    mel, mel_lens, al = eng.infer(tok, None, lens, max_step=1, dropout_seed=-1)
    assert mel.shape == (2, 80, 1) and mel_lens.cpu().tolist() == [1, 1]
    bad = dict(sd)
    del bad['decoder.gate_layer.weight']
    with pytest.raises(TtsAmdError, match='gate_layer'):
        Tacotron2Engine(bad, cfg, device=dev)
    with pytest.raises(TtsAmdError):
        eng.infer(torch.zeros(1, 2000, dtype=torch.long), None, None, max_step=2, dropout_seed=-1)


# ---- drop-in classes (reference models/tacotron2/networks.py) -------------------------------------

@pytest.fixture(scope='module')
def taco_ckpt(tmp_path_factory, synth_weights):
    import json
    import text
    from ttsamd.config import HIFIGAN_CONFIG
    d = tmp_path_factory.mktemp('taco')
    _, sd = _weights(gate_bias=-0.5, seed=4)   # a seed whose attention peaks on the separator after frame 0
    torch.save({'model': {k: torch.from_numpy(np.asarray(v).copy()) for k, v in sd.items()},
                'symbols': list(text.symbols)}, d / 'taco.pth')
    torch.save({'generator': {k: torch.from_numpy(v.copy()) for k, v in synth_weights['hifigan'].items()}}, d / 'hg.pth')
    with open(d / 'config.json', 'w') as f:
        json.dump(HIFIGAN_CONFIG, f)
    return str(d / 'taco.pth'), str(d / 'hg.pth'), str(d / 'config.json')


LINES = ["اَلسَّلامُ عَلَيكُم يَا صَدِيقِي", "صِفر", "أَربَعَة", "كِتَاب", "ثَلاثَة"]


def expected_cut(mel, ps_end):
    """The attention-peak cut as the reference's OUTPUT defines it (first frame at >= 80 % of the column's maximum, then the
    last kept frame three more times) -- written independently of the product's truncate_mel and itself pinned to the
    reference-generated golden by tests/test_taco_wrapper_golden.py::test_expected_cut_helper_is_the_reference."""
    ps = ps_end.detach().cpu().numpy()
    n_end = int(np.argmax(ps >= 0.8 * ps.max()))
    m = mel.detach().cpu().numpy()[:, :n_end]
    return torch.from_numpy(np.concatenate([m, np.repeat(m[:, -1:], 3, axis=1)], axis=1))


def test_dropin_tacotron2_ttmel(dev, taco_ckpt):
    """Tacotron2.ttmel: str / list, batched == single (dropout off), separator insertion and the
    attention-peak cut follow the reference (:125-152,155-206)."""
    import taco_oracle as T
    import text
    from models.tacotron2.networks import Tacotron2
    from text.symbols import SEPARATOR_TOKEN
    model = Tacotron2(taco_ckpt[0], n_symbol=len(text.symbols), decoder_max_step=48).to(dev)
    model.dropout_seed = -1
    mel1 = model.ttmel(LINES[0], postprocess_mel=False)
    assert mel1.dim() == 2 and mel1.shape[0] == 80 and mel1.device.type == 'cuda'
    # the same call restated with the oracle
    tokens = text.arabic_to_tokens(LINES[0])
    ids = torch.LongTensor(text.tokens_to_ids(tokens, model.phon_to_id))[None]
    ref, lr, _ = T.tacotron2_infer(model._sd, model.taco_config, ids, torch.zeros(1, dtype=torch.long), None,
                                   max_step=48, seed=-1)
    assert maxabs(mel1, ref[0]) < MEL_TOL
    # post-processing: "صِفر" ends in r -> separator inserted, mel cut + 3 replicated frames
    toks = text.arabic_to_tokens(LINES[1])
    assert toks[-3] == 'r'                       # not an open ending: reference golden test_needs_postprocessing_every_symbol
    toks.insert(-2, SEPARATOR_TOKEN)
    ids = torch.LongTensor(text.tokens_to_ids(toks, model.phon_to_id))[None]
    ref, _, al = T.tacotron2_infer(model._sd, model.taco_config, ids, torch.zeros(1, dtype=torch.long), None,
                                   max_step=48, seed=-1)
    want = expected_cut(ref[0], al[0, :, -3])
    got = model.ttmel(LINES[1])
    assert got.shape == want.shape and maxabs(got, want) < MEL_TOL
    assert torch.equal(got[:, -1], got[:, -4])
    # list input: one padded batch (the shorter utterances see the pad embedding in the encoder convs, as in
    # the reference) restated with the oracle on the same collated ids
    from models.tacotron2.networks import text_collate_fn
    prepared = [model._tokens_for(line, None, True) for line in LINES]
    ids_pad, lens_sorted, rev = text_collate_fn([torch.LongTensor(text.tokens_to_ids(t, model.phon_to_id))
                                                 for t, _ in prepared])
    ref, lr, al = T.tacotron2_infer(model._sd, model.taco_config, ids_pad, lens_sorted * 0, lens_sorted,
                                    max_step=48, seed=-1)
    batched = model.ttmel(LINES, batch_size=8)
    assert len(batched) == len(LINES)
    for i, j in enumerate(rev.tolist()):
        want = ref[j, :, :int(lr[j])]
        if prepared[i][1]:
            want = expected_cut(want, al[j, :int(lr[j]), int(lens_sorted[j]) - 3])
        assert batched[i].shape == want.shape and maxabs(batched[i], want) < MEL_TOL
    assert len(model.ttmel(LINES, batch_size=1)) == len(model.ttmel(LINES, batch_size=3)) == len(LINES)
    # speed resizes the time axis
    fast = model.ttmel(LINES[0], speed=1.25, postprocess_mel=False)
    assert fast.shape[1] == int(mel1.shape[1] / 1.25)


def test_dropin_tacotron2wave_tts(dev, taco_ckpt):
    from models.tacotron2 import Tacotron2Wave
    import text
    model = Tacotron2Wave(taco_ckpt[0], taco_ckpt[1], taco_ckpt[2], n_symbol=len(text.symbols)).to(dev)
    model.model.decoder_max_step = 40
    model.model.dropout_seed = 5
    w = model.tts(LINES[0], denoise=0)
    assert w.dim() == 1 and w.device.type == 'cpu' and w.numel() % 256 == 0 and bool(torch.isfinite(w).all())
    w2 = model.tts(LINES[0], denoise=0)
    assert torch.equal(w, w2)                                  # fixed seed -> reproducible
    model.model.dropout_seed = None                            # reference behaviour: new masks per call
    w3 = model.tts(LINES[0], denoise=0)
    assert w3.shape != w.shape or not torch.equal(w, w3)
    model.model.dropout_seed = -1
    # the one ragged vocoder launch of tts_batch == the reference's per-mel loop (:340-346)
    mels = model.model.ttmel_batch(LINES)
    batched = model.tts(LINES, batch_size=8, denoise=0.005)
    assert len(batched) == len(LINES)
    for mel, wav in zip(mels, batched):
        one = model.denoiser(model.vocoder(mel), 0.005)[0].cpu()
        assert one.shape == wav.shape and maxabs(one, wav) < 1e-4
    assert len(model.tts(LINES, batch_size=1, denoise=0)) == len(LINES)
    with pytest.raises(Exception):
        Tacotron2Wave(taco_ckpt[0], taco_ckpt[1], taco_ckpt[2], n_symbol=len(text.symbols)).tts(LINES[0])  # on the CPU


def test_inference_cli_both_models(dev, taco_ckpt, tmp_path, synth_weights):
    """inference.py twin (reference inference.py:20-64): --list file -> wavs/static<i>.wav at 22 050 Hz."""
    import inference
    import text
    from scipy.io import wavfile
    from ttsamd.config import NET_CONFIG
    lst = tmp_path / 'lines.txt'
    lst.write_text('\n'.join(LINES[:3]) + '\n', encoding='utf-8')
    fp = {k: torch.from_numpy(v.copy()) for k, v in synth_weights['fastpitch'].items()}
    torch.save({'model': fp, 'config': dict(NET_CONFIG), 'symbols': list(text.symbols)}, tmp_path / 'fp.pth')
    for model, ckpt in (('fastpitch', str(tmp_path / 'fp.pth')), ('tacotron2', taco_ckpt[0])):
        out = tmp_path / model
        inference.main(['--list', str(lst), '--model', model, '--checkpoint', ckpt, '--vocoder_sd', taco_ckpt[1],
                        '--vocoder_config', taco_ckpt[2], '--out_dir', str(out), '--batch_size', '2',
                        '--denoise', '0.005'])
        for i in range(3):
            sr, data = wavfile.read(out / 'wavs' / f'static{i}.wav')
            assert sr == 22050 and data.dtype == np.int16 and data.size > 0 and data.size % 256 == 0
        assert len((out / 'index.tsv').read_text(encoding='utf-8').splitlines()) == 3
    with pytest.raises(SystemExit):
        inference.main(['--cpu'])


@pytest.mark.parametrize('mode', ['1', '2'])
@pytest.mark.parametrize('num_speakers,B,L,steps,seed', [(40, 3, 23, 20, -1), (1, 8, 64, 24, 7), (40, 1, 7, 9, 0)])
def test_tacotron2_persistent_decoder_matches_oracle(dev, monkeypatch, num_speakers, B, L, steps, seed, mode, ttsopt):
    """TTSAMD_TACO_PERSISTENT=1 / 2: the whole decoder loop as one cooperative kernel (weights resident in LDS / registers,
    stop test on the device) — 1: six fence-free grid barriers per step, 2: no barriers, consumers poll the words they need
    (sentinel-filled arena) and the cells' big operands are folded in ahead — against the oracle at the same tolerances as
    the graph path, for both memory dims (640 multi-speaker, 512 single-speaker), with dropout on and off."""
    import taco_oracle as T
    from ttsamd.engine import Tacotron2Engine
    cfg, sd = _weights(gate_bias=-20.0, num_speakers=num_speakers)
    tok, lens = _tokens(B, L, 100 + B)
    sids = torch.arange(B) % num_speakers if num_speakers > 1 else None
    mel_ref, lens_ref, al_ref = T.tacotron2_infer(sd, cfg, tok, sids, lens, max_step=steps, seed=seed)
    eng = Tacotron2Engine(sd, cfg, device=dev)
    ttsopt.set('TTSAMD_TACO_PERSISTENT', mode)
    monkeypatch.setenv('TTSAMD_TACO_DEBUG', '1')
    mel, mel_lens, al = eng.infer(tok, sids, lens, max_step=steps, dropout_seed=seed)
    assert mel_lens.cpu().tolist() == np.asarray(lens_ref).tolist() == [steps] * B
    assert maxabs(al, al_ref) < ALIGN_TOL
    assert maxabs(mel, mel_ref) < MEL_TOL
    ttsopt.set('TTSAMD_TACO_PERSISTENT', '0')
    mel_g, _, al_g = eng.infer(tok, sids, lens, max_step=steps, dropout_seed=seed)       # and against the graph path
    assert maxabs(mel, mel_g) < MEL_TOL and maxabs(al, al_g) < ALIGN_TOL


@pytest.mark.parametrize('mode', ['1', '2'])
def test_tacotron2_persistent_decoder_stop_token(dev, monkeypatch, mode, ttsopt):
    """the persistent decoder's device-side stop test: utterances finish at different steps, the loop ends with the last"""
    import taco_oracle as T
    from ttsamd.engine import Tacotron2Engine
    cfg, sd = _weights(gate_bias=-20.0)
    tok, lens = _tokens(4, 19, 9)
    sids = torch.tensor([0, 3, 7, 39])
    stops = [12, 27, 5, 21]
    sd = _gate_for_stops(cfg, sd, tok, sids, lens, stops, max_step=40, seed=4)
    mel_ref, lens_ref, al_ref = T.tacotron2_infer(sd, cfg, tok, sids, lens, max_step=40, seed=4)
    ttsopt.set('TTSAMD_TACO_PERSISTENT', mode)
    mel, mel_lens, al = Tacotron2Engine(sd, cfg, device=dev).infer(tok, sids, lens, max_step=40, dropout_seed=4)
    assert mel_lens.cpu().tolist() == stops and mel.shape == mel_ref.shape
    assert maxabs(mel, mel_ref) < MEL_TOL and maxabs(al, al_ref) < ALIGN_TOL


@pytest.mark.parametrize('mode', ['1', '2'])
def test_tacotron2_persistent_decoder_segments(dev, monkeypatch, mode, ttsopt):
    """The persistent decoder as SEGMENTS of 8 steps, one cooperative launch each (the arena holds one segment; region 0 of a segment is
    the previous segment's last region, the per-thread cell states / cumulative attention / stop flags travel through the state buffer):
    utterances stop at steps 12 / 27 / 5 / 21 of 40, i.e. in segments 1 / 3 / 0 / 2, the loop ends inside segment 3 and the fifth
    launch never happens.  Same trajectory as the oracle, as the one-segment run and -- prenet dropout on -- bit for bit as itself."""
    import taco_oracle as T
    from ttsamd.engine import Tacotron2Engine
    cfg, sd = _weights(gate_bias=-20.0)
    tok, lens = _tokens(4, 19, 9)
    sids = torch.tensor([0, 3, 7, 39])
    stops = [12, 27, 5, 21]
    sd = _gate_for_stops(cfg, sd, tok, sids, lens, stops, max_step=40, seed=4)
    mel_ref, lens_ref, al_ref = T.tacotron2_infer(sd, cfg, tok, sids, lens, max_step=40, seed=4)
    eng = Tacotron2Engine(sd, cfg, device=dev)
    ttsopt.set('TTSAMD_TACO_PERSISTENT', mode)
    mel_one, lens_one, al_one = eng.infer(tok, sids, lens, max_step=40, dropout_seed=4)
    ttsopt.set('TTSAMD_TACO_SEG', '8')
    mel, mel_lens, al = eng.infer(tok, sids, lens, max_step=40, dropout_seed=4)
    mel2, _, _ = eng.infer(tok, sids, lens, max_step=40, dropout_seed=4)
    assert mel_lens.cpu().tolist() == stops == lens_one.cpu().tolist() and mel.shape == mel_ref.shape
    assert maxabs(mel, mel_ref) < MEL_TOL and maxabs(al, al_ref) < ALIGN_TOL
    assert torch.equal(mel, mel_one) and torch.equal(al, al_one) and torch.equal(mel, mel2)
    # workspace: one segment of regions, not max_step + 1 (the wrapper's decoder_max_step = 3000 was 0.5-0.9 GB)
    ttsopt.set('TTSAMD_TACO_SEG', None)
    nb = eng.lib.ttsamd_tacotron2_workspace_bytes(eng.handle, 8, 256, 3000)
    assert nb < 150 * (1 << 20) + 8 * 512 * 3000 * 4 * 2 + 64 * (1 << 20), nb     # arena + the two postnet buffers [8][512][3000] + the rest


def test_tacotron2_persistent_decoder_geometries(dev, monkeypatch, ttsopt):
    """The default (persistent, dataflow) decoder against the graph path over the corners of its residency plan: the largest
    token count (256: LDS 155-159 KB per block), both memory dims, odd batch sizes, one step, and back-to-back calls of
    different shapes on one engine (the exchange arena is re-filled with the sentinel per call)."""
    from ttsamd.engine import Tacotron2Engine
    for num_speakers in (40, 1):
        cfg, sd = _weights(gate_bias=-20.0, num_speakers=num_speakers)
        eng = Tacotron2Engine(sd, cfg, device=dev)
        for B, L, steps, seed in ((8, 256, 3, 3), (5, 100, 6, -1), (2, 33, 5, 1), (8, 256, 1, -1), (1, 1, 4, 2)):
            tok, lens = _tokens(B, L, 7 * B + L)
            sids = torch.arange(B) % num_speakers if num_speakers > 1 else None
            ttsopt.set('TTSAMD_TACO_PERSISTENT', '2')          # explicit: a geometry that did not fit or a time-out would raise
            mel, mel_lens, al = eng.infer(tok, sids, lens, max_step=steps, dropout_seed=seed)
            ttsopt.set('TTSAMD_TACO_PERSISTENT', '0')
            mel_g, lens_g, al_g = eng.infer(tok, sids, lens, max_step=steps, dropout_seed=seed)
            assert mel_lens.cpu().tolist() == lens_g.cpu().tolist() == [steps] * B
            assert bool(torch.isfinite(mel).all())
            assert maxabs(mel, mel_g) < MEL_TOL and maxabs(al, al_g) < ALIGN_TOL, (num_speakers, B, L, steps)


def test_tacotron2_persistent_explicit_request_raises_when_it_does_not_fit(dev, monkeypatch, ttsopt):
    """TTSAMD_TACO_PERSISTENT=1 / 2 is a demand, not a hint: a geometry outside the residency plan (batch 9 > 8) must raise instead
    of silently taking the graph path (so the geometry test above really proves the persistent kernel ran); unset, the same call
    runs on the graph path and matches it."""
    from ttsamd.engine import Tacotron2Engine
    from ttsamd.lib import TtsAmdError
    cfg, sd = _weights(gate_bias=-20.0)
    eng = Tacotron2Engine(sd, cfg, device=dev)
    tok, lens = _tokens(9, 12, 3)
    sids = torch.arange(9) % cfg['num_speakers']
    ttsopt.set('TTSAMD_TACO_PERSISTENT', '2')
    with pytest.raises(TtsAmdError, match='does not fit'):
        eng.infer(tok, sids, lens, max_step=4, dropout_seed=-1)
    ttsopt.set('TTSAMD_TACO_PERSISTENT', None)
    mel, mel_lens, _ = eng.infer(tok, sids, lens, max_step=4, dropout_seed=-1)
    ttsopt.set('TTSAMD_TACO_PERSISTENT', '0')
    mel_g, _, _ = eng.infer(tok, sids, lens, max_step=4, dropout_seed=-1)
    assert mel_lens.cpu().tolist() == [4] * 9 and maxabs(mel, mel_g) == 0.0


@pytest.mark.parametrize('mode', ['0', '2'])
def test_tacotron2_without_early_stopping(dev, monkeypatch, mode, ttsopt):
    """decoder_early_stopping=False (tacotron2_ms.py:139,169,197): the loop runs to max_step although every utterance's gate fired;
    the output has max_step frames, the per-utterance lengths are still the stop steps.  Graph path and persistent decoder vs the oracle."""
    import taco_oracle as T
    from ttsamd.engine import Tacotron2Engine
    cfg, sd = _weights(gate_bias=-20.0)
    tok, lens = _tokens(3, 15, 21)
    sids = torch.tensor([1, 5, 9])
    stops = [6, 11, 4]
    sd = _gate_for_stops(cfg, sd, tok, sids, lens, stops, max_step=20, seed=3)
    cfg = dict(cfg, decoder_early_stopping=False)
    mel_ref, lens_ref, al_ref = T.tacotron2_infer(sd, cfg, tok, sids, lens, max_step=20, seed=3)
    assert mel_ref.shape[2] == 20 and np.asarray(lens_ref).tolist() == stops
    ttsopt.set('TTSAMD_TACO_PERSISTENT', mode)
    mel, mel_lens, al = Tacotron2Engine(sd, cfg, device=dev).infer(tok, sids, lens, max_step=20, dropout_seed=3)
    assert mel.shape == (3, 80, 20) and mel_lens.cpu().tolist() == stops
    assert maxabs(mel, mel_ref) < MEL_TOL and maxabs(al, al_ref) < ALIGN_TOL
