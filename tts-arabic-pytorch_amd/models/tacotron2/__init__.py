"""Tacotron2 / Tacotron2Wave on the MI355X engine (reference models/tacotron2/networks.py:71-426,
BASELINE config 4).  `from models.tacotron2 import Tacotron2Wave` as in the reference README."""
from .networks import Tacotron2, Tacotron2Wave  # noqa: F401
from .tacotron2_ms import Tacotron2MS  # noqa: F401
