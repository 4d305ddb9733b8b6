"""`load_hifigan` with the reference's signature (vocoder/__init__.py:3-20)."""
import json

import torch


def load_hifigan(state_dict_path, config_file):
    from vocoder.hifigan.env import AttrDict
    from vocoder.hifigan.models import Generator

    with open(config_file) as f:
        h = AttrDict(json.loads(f.read()))
    generator = Generator(h)
    state_dict_g = torch.load(state_dict_path, map_location='cpu')
    generator.load_state_dict(state_dict_g['generator'])
    generator.eval()
    generator.remove_weight_norm()
    return generator
