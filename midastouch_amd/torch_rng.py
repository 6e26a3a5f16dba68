"""torch's CPU random stream, reproduced on the device (csrc/mt19937.hip, include/midas_hip.h midas_mt19937_*).

The reference resamples with `WeightedRandomSampler` -> `torch.multinomial(weights.double(), N, True)` on torch's default
CPU generator (modules/particle_filter.py:245): under `torch.manual_seed(s)` the N draws are the stream of
`torch.rand(N, dtype=float64)`.  `TorchCpuStream(seed)` keeps that generator's state in device memory and hands out the
same uniforms without a host generator or a per-frame upload; `skip_normal(numel)` steps over the outputs a
`torch.normal(..., size)` of float32 values would have taken (add_noise_to_odom, :326-335), so a caller that still draws
its motion noise on the host generator stays aligned with the reference's stream.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import _ptr


def normal_words(numel: int) -> int:
    """32-bit generator outputs one CPU torch.normal of `numel` float32 values consumes (ATen normal_fill: one per value,
    the last 16 drawn again when numel is not a multiple of 16).  Sizes below 16 take ATen's scalar path: not modelled."""
    if numel < 16:
        raise _lib.MidasError("torch.normal of fewer than 16 values uses another code path of ATen: not modelled")
    return numel + (16 if numel % 16 else 0)


class TorchCpuStream:
    def __init__(self, seed: int, device=None):
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.ctx = _lib.context(dev)
        self.device = self.ctx.device
        self.state = torch.zeros(626, dtype=torch.int32, device=self.device)
        self.pending_skip = 0
        self.manual_seed(seed)

    def manual_seed(self, seed: int):
        """torch.manual_seed(seed) for this stream."""
        self.ctx.bind_current_stream()
        self.ctx.call("midas_mt19937_seed", int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(self.state))
        self.pending_skip = 0
        return self

    def skip_words(self, n: int):
        """Step over n 32-bit outputs (applied with the next draw)."""
        self.pending_skip += int(n)
        return self

    def skip_normal(self, numel: int):
        """Step over what torch.normal(mean, std, size) with `numel` float32 values takes."""
        return self.skip_words(normal_words(numel))

    def rand64(self, N: int, out: torch.Tensor | None = None) -> torch.Tensor:
        """The next N values of torch.rand(N, dtype=torch.float64) (== the draws of torch.multinomial(w64, N, True))."""
        if out is None:
            out = torch.empty(int(N), dtype=torch.float64, device=self.device)
        self.ctx.bind_current_stream()
        self.ctx.call("midas_mt19937_rand64", _ptr(self.state), self.pending_skip, int(N), _ptr(out))
        self.pending_skip = 0
        return out
