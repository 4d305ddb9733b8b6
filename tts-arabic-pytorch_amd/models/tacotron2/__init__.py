"""Tacotron2 / Tacotron2Wave (reference models/tacotron2/networks.py:71-426, BASELINE config 4).

NOT BUILT in round 1: the model's arithmetic lives in the un-vendored
`torchaudio.models.tacotron2`, absent from the reference tree and from this image, and its
prenet dropout is stochastic at inference, so it cannot be pinned against the reference.
The classes exist so that `from models.tacotron2 import Tacotron2Wave` fails loudly at
construction instead of silently falling back to anything else."""
from ttsamd.lib import TtsAmdError


class _NotBuilt:
    def __init__(self, *args, **kwargs):
        raise TtsAmdError(f'{type(self).__name__}: the Tacotron2 path is not built on the MI355X engine yet '
                          '(DESIGN.md §7); use models.fastpitch.FastPitch2Wave')


class Tacotron2(_NotBuilt):
    pass


class Tacotron2Wave(_NotBuilt):
    pass
