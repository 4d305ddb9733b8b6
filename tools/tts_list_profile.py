"""Where the drop-in `FastPitch2Wave.tts(list, batch_size)` spends its host time (cProfile), synthetic checkpoints:
    python tools/tts_list_profile.py [batch_size] [n_lines]"""
import cProfile
import json
import os
import pstats
import sys
import tempfile
import time

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(R, 'tts-arabic-pytorch_amd'))
import torch  # noqa: E402


def main():
    import text
    from ttsamd import synth
    from ttsamd.config import NET_CONFIG, HIFIGAN_CONFIG
    from models.fastpitch import FastPitch2Wave
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    d = tempfile.mkdtemp()
    torch.save({'model': {k: torch.from_numpy(v.copy()) for k, v in synth.fastpitch_state_dict().items()}, 'config': dict(NET_CONFIG),
                'symbols': list(text.symbols)}, os.path.join(d, 'fp.pth'))
    torch.save({'generator': {k: torch.from_numpy(v.copy()) for k, v in synth.hifigan_state_dict().items()}}, os.path.join(d, 'hg.pth'))
    with open(os.path.join(d, 'config.json'), 'w') as f:
        json.dump(HIFIGAN_CONFIG, f)
    model = FastPitch2Wave(os.path.join(d, 'fp.pth'), vocoder_sd=os.path.join(d, 'hg.pth'), vocoder_config=os.path.join(d, 'config.json')).to('cuda:0')
    with open(os.path.join(R, 'tests', 'golden', 'infer_text_lines.json'), encoding='utf-8') as f:
        lines = json.load(f)[:n]
    for mode in ('0', '1'):
        os.environ['TTSAMD_TTS_PIPELINE'] = mode
        model.tts(lines[:2 * bs], batch_size=bs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.tts(lines, batch_size=bs)
        print(f'pipeline={mode} batch_size={bs}: {1e3 * (time.perf_counter() - t0):.1f} ms for {n} lines')
    os.environ['TTSAMD_TTS_PIPELINE'] = '0'
    pr = cProfile.Profile()
    pr.enable()
    model.tts(lines, batch_size=bs)
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(22)


if __name__ == '__main__':
    main()
