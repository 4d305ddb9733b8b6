"""Command line twin of the reference's inference.py (same flags and defaults, :96-111): synthesise every
line of --list with FastPitch2Wave / Tacotron2Wave on the MI355X and write <out_dir>/wavs/static<i>.wav.
The html sample page of the reference (utils/make_html.py) is out of scope; a plain index.tsv is written."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402

from utils import read_lines_from_file  # noqa: E402
from utils.audio import save_wav  # noqa: E402


def infer(args):
    if args.cpu or not torch.cuda.is_available():
        raise SystemExit('inference.py: this build runs on an MI355X only (no CPU path); drop --cpu')
    if args.model == 'fastpitch':
        from models.fastpitch import FastPitch2Wave as Model
    elif args.model == 'tacotron2':
        from models.tacotron2 import Tacotron2Wave as Model
    else:
        raise TypeError('model type not supported')
    model = Model(args.checkpoint, vocoder_sd=args.vocoder_sd, vocoder_config=args.vocoder_config).to('cuda').eval()
    os.makedirs(os.path.join(args.out_dir, 'wavs'), exist_ok=True)
    lines = [ln for ln in read_lines_from_file(args.list) if ln]
    idx = 0
    with open(os.path.join(args.out_dir, 'index.tsv'), 'w', encoding='utf-8') as index:
        for k in range(0, len(lines), args.batch_size):
            batch = lines[k:k + args.batch_size]
            wavs = model.tts(batch, batch_size=args.batch_size, denoise=args.denoise, speed=args.speed)
            for line, wav in zip(batch, wavs):
                save_wav(os.path.join(args.out_dir, 'wavs', f'static{idx}.wav'), wav, 22_050)
                index.write(f'wavs/static{idx}.wav\t{wav.numel()}\t{line}\n')
                idx += 1
    print(f'Saved files to: {args.out_dir}')


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--list', type=str, default='./data/infer_text.txt')
    p.add_argument('--model', type=str, default='fastpitch')
    p.add_argument('--checkpoint', type=str, default='pretrained/fastpitch_ar_adv.pth')
    p.add_argument('--vocoder_sd', type=str, default=None)
    p.add_argument('--vocoder_config', type=str, default=None)
    p.add_argument('--out_dir', type=str, default='samples/results')
    p.add_argument('--speed', type=float, default=1.0)
    p.add_argument('--denoise', type=float, default=0)
    p.add_argument('--batch_size', type=int, default=2)
    p.add_argument('--cpu', action='store_true')
    infer(p.parse_args(argv))


if __name__ == '__main__':
    main()
