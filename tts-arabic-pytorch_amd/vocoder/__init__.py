"""Vocoder loaders of the drop-in surface.

`load_hifigan(state_dict_path, config_file)` keeps the reference's signature
(vocoder/__init__.py:3): it returns a callable mel -> waveform module.  Here the module is
the HIP generator (vocoder.hifigan.models.Generator): the checkpoint's weight-normalised
tensors are handed to the C ABI as they are, which folds g*v/||v|| itself.
"""
import json


def _read_generator_checkpoint(path):
    import torch
    ckpt = torch.load(path, map_location='cpu')
    if 'generator' not in ckpt:
        raise KeyError(f"{path}: expected a HiFi-GAN checkpoint with a 'generator' entry")
    return ckpt['generator']


def load_hifigan(state_dict_path, config_file):
    from vocoder.hifigan.env import AttrDict
    from vocoder.hifigan.models import Generator

    with open(config_file, 'r') as fh:
        hparams = AttrDict(json.load(fh))
    vocoder = Generator(hparams, state_dict=_read_generator_checkpoint(state_dict_path))
    return vocoder.eval()
