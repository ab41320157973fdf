"""Mix2AffineQuantizer: fixed per-output-channel integer bit-widths supplied as a tensor.  Same
constructor and attributes as the reference class (quantize/quantizer_mix_v2.py:29-169); the
arithmetic is the uniform row-wise kernel with a per-row bit-width and a clamped asymmetric scale."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import functional as F
from ._lib import MiqError
from .quantizer import lwc_clip_factors

CLIPMIN = 1e-5


class Mix2AffineQuantizer(nn.Module):
    def __init__(
        self,
        target_bit: int = 4,
        symmetric: bool = False,
        n_bits: torch.Tensor = [],
        per_channel_axes=[],
        metric="minmax",
        dynamic=False,
        dynamic_method="per_channel",
        group_size=None,
        shape=None,
        mix2_nbits: torch.Tensor = [],
        lwc=False,
    ):
        super().__init__()
        self.symmetric = symmetric
        self.target_bit = target_bit
        self.n_bits = mix2_nbits.unsqueeze(dim=-1)
        self.q_min = torch.zeros_like(self.n_bits)
        self.q_max = 2 ** (self.n_bits) - 1
        self.per_channel_axes = per_channel_axes
        self.metric = metric
        self.cluster_counts = None
        self.cluster_dim = None
        self.scale = None
        self.zero_point = None
        self.round_zero_point = None
        self.cached_xmin = None
        self.cached_xmax = None
        self.dynamic = dynamic
        self.dynamic_method = dynamic_method
        self.deficiency = 0
        self.lwc = lwc
        self.mode = "calibration"
        init_value = 4.0
        if lwc:
            if group_size:
                dim1 = int(shape[0] * math.ceil(shape[1] / group_size))
                self.deficiency = shape[-1] % group_size
                if self.deficiency > 0:
                    self.deficiency = group_size - self.deficiency
                    assert self.symmetric
            else:
                dim1 = shape[0]
            self.upbound_factor = nn.Parameter(torch.ones((dim1, 1)) * init_value)
            self.lowbound_factor = nn.Parameter(torch.ones((dim1, 1)) * init_value)
        self.sigmoid = nn.Sigmoid()
        self.enable = True
        self.group_size = group_size
        self._bits_dev = None

    def _row_bits(self, x):
        b = self._bits_dev
        if b is None or b.device != x.device:
            b = self._bits_dev = self.n_bits.to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
        return b

    def forward(self, x: torch.Tensor):
        self.shape = x.shape
        if self.group_size:
            raise MiqError("per-row bit-widths cannot be combined with group_size (one bit-width per output row)")
        if x.dtype != torch.float32:
            raise MiqError("the fixed mixed-bit quantizer runs in float32 (its bit tensor is float32 in the reference)")
        up, low = lwc_clip_factors(self, x)
        out, scale, rzp = F.UniformRowFQ.apply(x, up, low, 0, bool(self.symmetric), 0, self._row_bits(x), True)
        return out

    def register_scales_and_zeros(self):
        self.register_buffer("scales", self.scale)
        self.register_buffer("zeros", self.round_zero_point)
        del self.scale
        del self.round_zero_point

    def free(self):
        self.cached_xmax = None
        self.cached_xmin = None
