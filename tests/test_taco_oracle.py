"""CPU checks of oracle/taco_oracle.py (Tacotron2 restatement, PARITY UNPINNED — see its header):
internal consistency only, since neither torchaudio nor a Tacotron2 golden exists in the reference."""
import numpy as np
import torch


def _setup(B=2, L=9, gate_bias=-20.0):
    from ttsamd.config import TACOTRON2_CONFIG
    from ttsamd.synth import tacotron2_state_dict
    cfg = dict(TACOTRON2_CONFIG)
    sd = tacotron2_state_dict(cfg, seed=0, gate_bias=gate_bias)
    g = torch.Generator().manual_seed(3)
    tok = torch.randint(1, 40, (B, L), generator=g)
    lens = torch.tensor([L, L - 3][:B])
    tok = tok * (torch.arange(L)[None] < lens[:, None])
    return cfg, sd, tok, lens


def test_keep_mask_is_a_fair_deterministic_coin():
    import taco_oracle as T
    a = T.keep_mask(7, 0, 3, 4, 256)
    assert a.shape == (4, 256) and set(np.unique(a.numpy()).tolist()) <= {0.0, 2.0}
    assert torch.equal(a, T.keep_mask(7, 0, 3, 4, 256))
    assert not torch.equal(a, T.keep_mask(7, 1, 3, 4, 256)) and not torch.equal(a, T.keep_mask(8, 0, 3, 4, 256))
    big = torch.cat([T.keep_mask(1, 0, s, 8, 256) for s in range(64)])
    assert abs(float(big.mean()) - 1.0) < 0.02            # E[mask] = 1 (p = 0.5, scale 2)


def test_oracle_shapes_lengths_and_attention_rows():
    import taco_oracle as T
    cfg, sd, tok, lens = _setup()
    mel, mel_lens, al = T.tacotron2_infer(sd, cfg, tok, torch.tensor([0, 5]), lens, max_step=6, seed=-1)
    assert mel.shape == (2, 80, 6) and al.shape == (2, 6, 9) and mel_lens.tolist() == [6, 6]
    assert torch.allclose(al.sum(-1), torch.ones(2, 6), atol=1e-5)
    assert float(al[1, :, 6:].abs().max()) == 0.0          # padded tokens get no attention
    # utterance 0 is unaffected by its batch mates in the decoder; the encoder convs see the pad embedding
    # only next to shorter utterances, so the longest one equals its unbatched run
    mel0, _, al0 = T.tacotron2_infer(sd, cfg, tok[:1], torch.tensor([0]), lens[:1], max_step=6, seed=-1)
    assert float((mel0 - mel[:1]).abs().max()) < 1e-4 and float((al0 - al[:1]).abs().max()) < 1e-5


def test_oracle_gate_stops_and_counts_like_the_reference_loop():
    import taco_oracle as T
    cfg, sd, tok, lens = _setup(gate_bias=20.0)             # sigmoid(gate) > 0.5 at the first step
    mel, mel_lens, al = T.tacotron2_infer(sd, cfg, tok, None, lens, max_step=6, seed=-1)
    assert mel.shape == (2, 80, 1) and mel_lens.tolist() == [1, 1]
    a = T.tacotron2_infer(sd, cfg, tok, None, lens, max_step=3, seed=5)[0]
    b = T.tacotron2_infer(sd, cfg, tok, None, lens, max_step=3, seed=5)[0]
    assert torch.equal(a, b)


def test_oracle_blocks_equal_the_torch_modules_torchaudio_builds_on():
    """torchaudio's _Encoder is nn.LSTM(bidirectional) on a packed sequence, its decoder cells are nn.LSTMCell:
    the oracle's hand-written recurrences must equal those torch modules on the same weights (op-level pin;
    the wiring between the blocks stays a restatement)."""
    import taco_oracle as T
    cfg, sd, tok, lens = _setup(B=2, L=9)
    tr = {}
    T.tacotron2_infer(sd, cfg, tok, torch.tensor([0, 5]), lens, max_step=1, seed=-1, trace=tr)
    lstm = torch.nn.LSTM(512, 256, 1, batch_first=True, bidirectional=True)
    lstm.load_state_dict({k[len('encoder.lstm.'):]: torch.from_numpy(v) for k, v in sd.items()
                          if k.startswith('encoder.lstm.')})
    packed = torch.nn.utils.rnn.pack_padded_sequence(tr['conv_out'], lens, batch_first=True)
    with torch.no_grad():
        out, _ = torch.nn.utils.rnn.pad_packed_sequence(lstm(packed)[0], batch_first=True)
    assert float((out - tr['enc']).abs().max()) < 2e-6
    cell = torch.nn.LSTMCell(896, 1024)
    cell.load_state_dict({k[len('decoder.attention_rnn.'):]: torch.from_numpy(v) for k, v in sd.items()
                          if k.startswith('decoder.attention_rnn.')})
    g = torch.Generator().manual_seed(0)
    x, h, c = torch.randn(2, 896, generator=g), torch.randn(2, 1024, generator=g), torch.randn(2, 1024, generator=g)
    W = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    with torch.no_grad():
        h1, c1 = cell(x, (h, c))
    h2, c2 = T._lstm_cell(x, h, c, W, 'decoder.attention_rnn')
    assert float((h1 - h2).abs().max()) < 2e-6 and float((c1 - c2).abs().max()) < 2e-6
