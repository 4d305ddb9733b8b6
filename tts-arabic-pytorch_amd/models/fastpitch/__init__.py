from ttsamd.config import NET_CONFIG as net_config  # noqa: F401  (models/fastpitch/__init__.py:3-41)

from .networks import FastPitch, FastPitch2Wave  # noqa: F401,E402
