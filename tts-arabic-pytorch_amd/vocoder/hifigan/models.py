"""Drop-in for the reference's vocoder.hifigan.models.Generator (vocoder/hifigan/models.py:86-136):
an nn.Module whose forward is the hand-written HIP generator behind the C ABI."""
import numpy as np
import torch
import torch.nn as nn

from ttsamd.engine import HifiGanEngine
from ttsamd.lib import TtsAmdError

LRELU_SLOPE = 0.1


class _HipModule(nn.Module):
    """Weights live in C-ABI handles, one per device, created lazily and kept; `.to()/.cuda()/.cpu()`
    move a 1-element anchor parameter that tracks the device the next call runs on.  Running on
    a non-ROCm device raises: there is no CPU path."""

    def __init__(self):
        super().__init__()
        self._anchor = nn.Parameter(torch.zeros(1), requires_grad=False)
        self._engines = {}

    @property
    def device(self):
        return self._anchor.device

    # Device handles survive `.cpu()` / `.to(other)`: the reference's server moves each model to the GPU and back on every request
    # (utils/app_utils.py:65,81) to share a small card; with 288 GB of HBM the packed weights (< 1 GB for every model of this repo
    # together) simply stay resident, so the next `.to('cuda')` costs nothing.  `release_device_memory()` frees them explicitly.

    def release_device_memory(self):
        """Destroy the C-ABI handles (device copies of the packed weights) of every device."""
        self._engines.clear()

    def _engine(self, factory):
        dev = self._anchor.device
        if dev.type != 'cuda':
            raise TtsAmdError(f'{type(self).__name__} is on {dev}: the MI355X path has no CPU fallback; '
                              'move the module with .to("cuda")')
        key = str(dev)
        if key not in self._engines:
            self._engines[key] = factory(dev)
        return self._engines[key]


class Generator(_HipModule):
    def __init__(self, h, state_dict=None):
        super().__init__()
        self.h = h
        self.num_kernels = len(h['resblock_kernel_sizes'])
        self.num_upsamples = len(h['upsample_rates'])
        self._sd = None
        if state_dict is not None:
            self.load_state_dict(state_dict)

    def load_state_dict(self, state_dict, strict=True):
        self._sd = {k: (v.detach().cpu().float().numpy() if hasattr(v, 'detach') else np.asarray(v, np.float32))
                    for k, v in state_dict.items()}
        self._engines.clear()

    def state_dict(self, *a, **k):
        return {k_: torch.from_numpy(v) for k_, v in (self._sd or {}).items()}

    def remove_weight_norm(self):
        """No-op: the C-ABI loader folds g*v/||v|| itself (ttsamd_hifigan_create)."""

    def engine(self):
        if self._sd is None:
            raise TtsAmdError('Generator has no weights: call load_state_dict first')
        return self._engine(lambda dev: HifiGanEngine(self._sd, dict(self.h), device=dev))

    @torch.inference_mode()
    def forward(self, x, lens=None):
        """x [80,T] -> [1,256T] (unbatched, models/fastpitch/networks.py:312,341) or
        [B,80,T] -> [B,1,256T] (test.py:62-63).  `lens` (int64 [B], extension) makes the batch ragged."""
        eng = self.engine()
        if x.dim() == 2:
            return eng.forward(x[None])
        return eng.forward(x, lens)[:, None, :]
