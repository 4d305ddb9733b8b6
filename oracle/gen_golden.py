"""Generate tests/golden/*.npz by running the REAL reference (/root/reference) in this
container.  Test infrastructure; run manually:  python oracle/gen_golden.py

The reference modules are instantiated exactly as its own loaders do
(models/fastpitch/networks.py:45-75,257-287; vocoder/__init__.py:3-20), fed with the
deterministic synthetic weights of ttsamd.synth (the tree ships no weights), and their
inputs/outputs are stored as small fixtures.  No reference source is copied; the GPU
box only ever sees the .npz data files.
"""
import hashlib
import importlib.util
import json
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, 'tests', 'golden')


def _load_pkg_module(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REPO, 'tts-arabic-pytorch_amd', *rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


# our own synth generator, loaded by path (package names must not shadow the reference's)
import types
_pkg = types.ModuleType('ttsamd'); _pkg.__path__ = [os.path.join(REPO, 'tts-arabic-pytorch_amd', 'ttsamd')]
sys.modules['ttsamd'] = _pkg
config = _load_pkg_module('ttsamd.config', ('ttsamd', 'config.py'))
synth = _load_pkg_module('ttsamd.synth', ('ttsamd', 'synth.py'))

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _refstub
_refstub.install()

import torch  # noqa: E402
import text as ref_text  # noqa: E402  (reference text front-end)
from models.fastpitch import net_config as ref_net_config  # noqa: E402
from models.fastpitch.fastpitch.model import FastPitch as RefFastPitchCore  # noqa: E402
from models.fastpitch.fastpitch.model import regulate_len as ref_regulate_len  # noqa: E402
from models.fastpitch.networks import FastPitch2Wave, text_collate_fn  # noqa: E402
from vocoder import load_hifigan  # noqa: E402
from vocoder.hifigan.env import AttrDict  # noqa: E402
from vocoder.hifigan.models import Generator  # noqa: E402

torch.manual_seed(0)
torch.set_grad_enabled(False)


def sd_digest(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k]).tobytes())
    return h.hexdigest()


def t(sd):
    return {k: torch.from_numpy(v.copy()) for k, v in sd.items()}


def save(name, **arrs):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **{k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                                 for k, v in arrs.items()})
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} kB')


def build_ref_fastpitch(cfg, seed=0):
    assert ref_net_config == config.NET_CONFIG, 'NET_CONFIG drifted from the reference'
    m = RefFastPitchCore(**cfg)
    sd = synth.fastpitch_state_dict(cfg, seed)
    missing, unexpected = m.load_state_dict(t(sd), strict=False)
    assert all(k.startswith('attention.') for k in missing), missing
    assert not unexpected, unexpected
    return m.eval(), sd


def build_ref_hifigan(seed=0):
    with open('pretrained/hifigan-asc-v1/config.json') as f:
        h = json.load(f)
    for k, v in config.HIFIGAN_CONFIG.items():
        assert h[k] == v, (k, h[k], v)
    sd = synth.hifigan_state_dict(None, seed)
    with tempfile.NamedTemporaryFile(suffix='.pth', delete=False) as f:
        torch.save({'generator': t(sd)}, f.name)
    with torch.enable_grad():        # remove_parametrizations leaves plain tensors under no_grad
        g = load_hifigan(f.name, 'pretrained/hifigan-asc-v1/config.json')
    os.unlink(f.name)
    return g, sd


def main():
    os.makedirs(OUT, exist_ok=True)
    digests = {}

    # ---------------- HiFi-GAN ----------------
    g, hsd = build_ref_hifigan()
    digests['hifigan_seed0'] = sd_digest(hsd)
    # weight-norm fold fixture (g, v -> w) for one Conv1d and one ConvTranspose1d
    gsd = g.state_dict()
    save('weightnorm_fold',
         conv_g=hsd['resblocks.3.convs1.1.parametrizations.weight.original0'],
         conv_v=hsd['resblocks.3.convs1.1.parametrizations.weight.original1'],
         conv_w=gsd['resblocks.3.convs1.1.weight'],
         convt_g=hsd['ups.2.parametrizations.weight.original0'],
         convt_v=hsd['ups.2.parametrizations.weight.original1'],
         convt_w=gsd['ups.2.weight'])
    rng = np.random.default_rng(7)
    for T in (1, 7, 40):
        mel = torch.from_numpy((rng.standard_normal((80, T)) * 1.5 - 4.0).astype(np.float32))
        stages = {}
        if T == 7:
            hooks = []
            hooks.append(g.conv_pre.register_forward_hook(lambda m, i, o: stages.__setitem__('conv_pre', o.clone())))
            for i in range(4):
                hooks.append(g.ups[i].register_forward_hook(
                    lambda m, i_, o, i=i: stages.__setitem__(f'ups{i}', o.clone())))
                hooks.append(g.resblocks[3 * i].register_forward_hook(
                    lambda m, i_, o, i=i: stages.__setitem__(f'rb{i}_0', o.clone())))
            hooks.append(g.conv_post.register_forward_hook(lambda m, i_, o: stages.__setitem__('post_in', i_[0].clone())))
        wave = g(mel)                                 # 2-D in, as networks.py:312,341
        wave3 = g(mel[None])                          # 3-D in, as test.py:62-63
        assert torch.equal(wave3[0], wave) or (wave3[0] - wave).abs().max() < 1e-6
        if T == 7:
            for h_ in hooks:
                h_.remove()
        save(f'hifigan_T{T}', mel=mel, wave=wave, **{'stage_' + k: v for k, v in stages.items()})
    print('wave abs max', float(wave.abs().max()), 'std', float(wave.std()))

    # ---------------- regulate_len ----------------
    r = np.random.default_rng(11)
    dur = (r.random((4, 13)) * 9.0).astype(np.float32)
    dur[1, 9:] = 0.0
    dur[3, :] = np.array([0.49999997, 0.5, 1.5, 2.5, 0.0, 3.4999998, 74.5, 75.0, 0.0, 1.0, 2.0, 0.25, 7.75], np.float32)
    enc = np.zeros((4, 13, 2), np.float32)
    enc[:, :, 0] = np.arange(1, 14)[None]
    enc[:, :, 1] = r.standard_normal((4, 13))
    reg = {'dur': dur, 'enc': enc}
    for pace in (0.8, 1.0, 1.25):
        rep, dl = ref_regulate_len(torch.from_numpy(dur), torch.from_numpy(enc), pace)
        tag = str(pace).replace('.', 'p')
        reg[f'idx_{tag}'] = (rep[:, :, 0].round().to(torch.int32) - 1)        # -1 = zero row
        reg[f'dec_lens_{tag}'] = dl
        reg[f'rep1_{tag}'] = rep[:, :, 1]
    save('regulate_len', **reg)

    # ---------------- FastPitch.infer: ragged padded batch, forced durations ----------------
    fp, fsd = build_ref_fastpitch(dict(config.NET_CONFIG))
    digests['fastpitch_seed0'] = sd_digest(fsd)
    r = np.random.default_rng(5)
    lens = [16, 9, 5]
    ids = np.zeros((3, 16), np.int64)
    for b, n in enumerate(lens):
        ids[b, :n] = 1 + r.integers(0, 39, n)
    dur_tgt = (1 + r.integers(0, 5, ids.shape)).astype(np.float32) * (ids != 0)
    trace = {}
    hooks = []
    for part in ('encoder', 'decoder'):
        lyr = getattr(fp, part).layers[0]
        hooks.append(lyr.dec_attn.register_forward_hook(
            lambda m, i, o, part=part: trace.__setitem__(part + '_l0_attn', o.clone())))
        hooks.append(lyr.pos_ff.register_forward_hook(
            lambda m, i, o, part=part: trace.__setitem__(part + '_l0_ff', o.clone())))
        hooks.append(getattr(fp, part).register_forward_hook(
            lambda m, i, o, part=part: trace.__setitem__(part + '_out', o[0].clone())))
    mel, dec_lens, dur_pred, pitch_pred, energy_pred = fp.infer(
        torch.from_numpy(ids), dur_tgt=torch.from_numpy(dur_tgt))
    for h_ in hooks:
        h_.remove()
    save('fastpitch_b3_durtgt', ids=ids, dur_tgt=dur_tgt, mel=mel, dec_lens=dec_lens,
         dur_pred=dur_pred, pitch_pred=pitch_pred, energy_pred=energy_pred, **trace)
    print('mel abs max', float(mel.abs().max()), 'dur_pred mean', float(dur_pred[ids != 0].mean()))

    # predicted durations, pace and pitch transform (B=2; seeds chosen so that no repeat
    # count sits within 1e-3 of a rounding boundary)
    from models.fastpitch.networks import pitch_trf
    for tag, pace, ptr in (('p1', 1.0, None), ('p0p9_pitch', 0.9, pitch_trf(1.3, 0.2))):
        seed = 21
        while True:
            r = np.random.default_rng(seed)
            ids2 = np.zeros((2, 12), np.int64)
            ids2[0, :12] = 1 + r.integers(0, 39, 12)
            ids2[1, :7] = 1 + r.integers(0, 39, 7)
            out = fp.infer(torch.from_numpy(ids2), pace=pace, pitch_transform=ptr)
            frac = (out[2] / pace + 0.5) % 1.0
            margin = float(torch.minimum(frac, 1 - frac)[torch.from_numpy(ids2 != 0)].min())
            if margin > 5e-3:
                break
            seed += 1
        save(f'fastpitch_b2_pred_{tag}', ids=ids2, pace=np.float32(pace),
             pitch_mul=np.float32(1.3 if ptr else 1.0), pitch_add=np.float32(0.2 if ptr else 0.0),
             mel=out[0], dec_lens=out[1], dur_pred=out[2], pitch_pred=out[3], energy_pred=out[4],
             margin=np.float32(margin))
        print(tag, 'seed', seed, 'margin', margin, 'dec_lens', out[1].tolist())

    # multi-speaker (config 5 front half): n_speakers=4, speaker 2
    cfg4 = dict(config.NET_CONFIG, n_speakers=4, speaker_emb_weight=1.0)
    fp4, fsd4 = build_ref_fastpitch(cfg4)
    digests['fastpitch_spk4_seed0'] = sd_digest(fsd4)
    out = fp4.infer(torch.from_numpy(ids), dur_tgt=torch.from_numpy(dur_tgt), speaker=2)
    save('fastpitch_b3_spk2', ids=ids, dur_tgt=dur_tgt, mel=out[0], dec_lens=out[1],
         dur_pred=out[2], pitch_pred=out[3], energy_pred=out[4])

    # ---------------- text front-end + collate (config-1 inputs; §8 f1) ----------------
    with open('data/infer_text.txt', encoding='utf-8') as f:
        lines = [ln.strip() for ln in f.read().splitlines() if ln.strip()]
    tok_ids = [np.asarray(ref_text.tokens_to_ids(ref_text.arabic_to_tokens(ln, append_space=False)), np.int64)
               for ln in lines]
    flat = np.concatenate(tok_ids)
    offs = np.cumsum([0] + [len(x) for x in tok_ids]).astype(np.int64)
    padded, lens_sorted, rev = text_collate_fn([torch.from_numpy(x) for x in tok_ids[:5]])
    save('infer_text_ids', flat=flat, offsets=offs, symbols=np.array(ref_text.symbols),
         collate5_padded=padded, collate5_lens=lens_sorted, collate5_rev=rev)
    with open(os.path.join(OUT, 'infer_text_lines.json'), 'w', encoding='utf-8') as f:
        json.dump(lines, f, ensure_ascii=False, indent=0)
    print('infer_text: lines', len(lines), 'tokens', offs[-1], 'min/max', min(map(len, tok_ids)), max(map(len, tok_ids)))

    # ---------------- end-to-end FastPitch2Wave.tts ----------------
    with tempfile.TemporaryDirectory() as td:
        fpath, hpath = os.path.join(td, 'fp.pth'), os.path.join(td, 'hg.pth')
        # the reference loader is strict (networks.py:60): complete the dict with the
        # training-only `attention.*` tensors of a fresh reference module (unused by infer)
        full = {k: v for k, v in fp.state_dict().items() if k.startswith('attention.')}
        full.update(t(fsd))
        torch.save({'model': full, 'config': dict(config.NET_CONFIG), 'symbols': list(ref_text.symbols)}, fpath)
        torch.save({'generator': t(hsd)}, hpath)
        with torch.enable_grad():
            model = FastPitch2Wave(fpath, vocoder_sd=hpath, vocoder_config='pretrained/hifigan-asc-v1/config.json')
    order = np.argsort([len(x) for x in tok_ids])
    pick = [int(order[0]), int(order[1]), int(order[2])]
    texts = [lines[i] for i in pick]
    waves0 = model.tts(texts, batch_size=3, denoise=0.0)
    waves_d = model.tts(texts[:2], batch_size=2, denoise=0.005)
    single, mel_single = model.tts(texts[0], denoise=0.0, return_mel=True)
    e2e = {'line_idx': np.asarray(pick, np.int64), 'bias_spec': model.denoiser.bias_spec,
           'single_wave': single, 'single_mel': mel_single}
    for i, wv in enumerate(waves0):
        e2e[f'wave{i}'] = wv
    for i, wv in enumerate(waves_d):
        e2e[f'wave_dn{i}'] = wv
    save('e2e_tts', **e2e)
    print('e2e lens', [len(w) for w in waves0], 'amp', [float(w.abs().max()) for w in waves0])

    # ---------------- MelVocos('22k') (config 5 back half) ----------------
    from vocoder.vocos.pretrained import MelVocos
    with torch.enable_grad():
        mv = MelVocos('22k')
    vsd = synth.vocos_state_dict()
    digests['vocos_seed0'] = sd_digest(vsd)
    full = {k: v for k, v in mv.state_dict().items() if k not in vsd}      # window / feature-extractor buffers
    full.update(t(vsd))
    mv.load_state_dict(full)                                               # post-hook recomputes bias_vec
    mv.eval()
    rngv = np.random.default_rng(13)
    vg = {'bias_vec': mv.bias_vec}
    for T in (1, 5, 33):
        melv = torch.from_numpy((rngv.standard_normal((2, 80, T)) * 1.5 - 4.0).astype(np.float32))
        vg[f'mel_T{T}'] = melv
        vg[f'wave_T{T}'] = mv(melv)
        vg[f'wave_dn_T{T}'] = mv(melv, denoise=0.3)
    save('vocos_22k', **vg)
    print('vocos wave amp', float(vg['wave_T33'].abs().max()), 'bias max', float(mv.bias_vec.max()))

    with open(os.path.join(OUT, 'digests.json'), 'w') as f:
        json.dump(digests, f, indent=1)


if __name__ == '__main__':
    main()
