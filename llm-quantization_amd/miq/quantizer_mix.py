"""MixAffineQuantizer: learnable per-output-channel bit-width (blend of the floor/ceil bit grids).
Same constructor, Parameters (``n_bits`` [O,1], ``upbound_factor``/``lowbound_factor``) and methods as
the reference class (quantize/quantizer_mix.py:36-230), computed by libmiq (float32)."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import functional as F
from ._lib import MiqError
from .quantizer import lwc_clip_factors

CLIPMIN = 1e-5


def round_ste(x: torch.Tensor):
    return (x.round() - x).detach() + x


def floor_ste(x: torch.Tensor):
    return (x.floor() - x).detach() + x


def ceil_ste(x: torch.Tensor):
    return (x.ceil() - x).detach() + x


class MixAffineQuantizer(nn.Module):
    def __init__(
        self,
        target_bit: int = 4,
        n_bits: torch.Tensor = [],
        hard_assign=False,
        threshold=0.0,
        symmetric: bool = False,
        per_channel_axes=[],
        metric="minmax",
        dynamic=False,
        dynamic_method="per_channel",
        group_size=None,
        shape=None,
        lwc=False,
    ):
        super().__init__()
        self.symmetric = symmetric
        self.target_bit = target_bit
        if isinstance(n_bits, (list, tuple)) and len(n_bits) == 0:
            rows = shape[0]
            # reference init: target + 0.5 + U(0, 0.25) (it also moves the tensor to "cuda" there;
            # here the module is moved with .to(device) like any other)
            n_bits = torch.ones(rows, 1) * self.target_bit + 0.5 + torch.rand(rows, 1) / 4
        self.n_bits = nn.Parameter(n_bits)
        self.hard_assign = hard_assign
        self.threshold = threshold
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

    def forward(self, x: torch.Tensor):
        self.shape = x.shape
        if isinstance(self.n_bits, nn.Parameter):
            self.n_bits.data = self.n_bits.data.clamp(2, 8)  # keeps the grids valid (reference :145)
        n_bits = self.n_bits
        if x.dtype != torch.float32:
            raise MiqError("the learnable mixed-bit quantizer runs in float32 (the calibration loop calls qlayer.float())")
        up, low = lwc_clip_factors(self, x)
        return F.MixRowFQ.apply(x, n_bits, up, low, bool(self.symmetric), int(self.group_size or 0),
                                bool(self.hard_assign), float(self.threshold))

    def register_scales_and_zeros(self):
        self.register_buffer("scales", self.scale)
        self.register_buffer("zeros", self.round_zero_point)
        del self.scale
        del self.round_zero_point

    def free(self):
        self.cached_xmax = None
        self.cached_xmin = None

    def model_size_loss(self):
        """(mean bits per weight, target bits) as 0-dim tensors; the caller applies the MSE."""
        prop = self.shape[0] * self.shape[1]
        n_bits = self.n_bits
        if self.hard_assign:
            n_bits = round_ste(n_bits + self.threshold)
        self.target_size = self.shape[0] * self.shape[1] * self.target_bit
        self.target_size = torch.tensor(self.target_size).to(n_bits.device) / prop
        self.model_size = torch.sum(n_bits * self.shape[1]) / prop
        return self.model_size, self.target_size

    def set_hard(self):
        self.hard_assign = True

    def round_nbits(self):
        """Replace the Parameter by its rounded values (a plain tensor from then on)."""
        self.n_bits_round = self.n_bits.data
        self.n_bits_round = round_ste(self.n_bits_round + self.threshold)
        del self.n_bits
        self.n_bits = self.n_bits_round
