"""Drop-in for the reference's models/diacritizers: `load_vowelizer(name, config)` returns a
Shakkelha / Shakkala tagger whose network runs in libttsamd (csrc/tagger.hip)."""
from .shakkala.network import Shakkala
from .shakkelha.network import Shakkelha


def load_vowelizer(name: str, config):
    if name == 'shakkala':
        return Shakkala(sd_path=config.shakkala_path)
    if name == 'shakkelha':
        return Shakkelha(sd_path=config.shakkelha_path)
    raise ValueError(f"unknown vowelizer '{name}' (options: 'shakkala', 'shakkelha')")
