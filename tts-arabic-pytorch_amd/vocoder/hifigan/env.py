class AttrDict(dict):
    """dict with attribute access (vocoder/hifigan/env.py:5-8)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__ = self
