"""ExpAffineQuantizer (v2: power-of-two levels with a zero level): same constructor and attribute
surface as the reference class (quantize/quantizer_exp_v2.py:108-210), computed by libmiq.
This is the exponential quantizer QuantLinear instantiates for ``weight_exp_quant``."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import functional as F
from ._lib import MiqError
from .quantizer import lwc_clip_factors

CLIPMIN = 1e-5


class ExpAffineQuantizer(nn.Module):
    def __init__(
        self,
        n_bits: int = 8,
        symmetric: bool = True,
        per_channel_axes=[],
        metric="minmax",
        dynamic=False,
        dynamic_method="per_cluster",
        group_size=None,
        shape=None,
        lwc=False,
        target_bit=0,
    ):
        super().__init__()
        self.symmetric = symmetric
        self.n_bits = n_bits
        if symmetric:
            self.qmin = -1 << ((1 << self.n_bits - 1) - 2)
            self.qmax = 1 << ((1 << self.n_bits - 1) - 2)
        else:
            raise MiqError("the exponential quantizer only supports symmetric quantization")
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

    def change_n_bits(self, n_bits):
        # The reference sets qmin = 0 here (quantizer_exp_v2.py:169-172); the kernels derive the level
        # range from n_bits, so only the attribute mirrors that quirk.
        self.n_bits = n_bits
        self.qmin = 0
        self.qmax = 1 << ((1 << self.n_bits - 1) - 2)

    def forward(self, x: torch.Tensor):
        if not 2 <= self.n_bits <= 4:
            raise MiqError(f"exponential quantizer supports 2..4 bits, got {self.n_bits}")
        up, low = lwc_clip_factors(self, x)
        out, scale = F.Exp2RowFQ.apply(x, up, low, self.n_bits, int(self.group_size or 0))
        self.scale = scale
        self.zero_point = None
        # the reference traps into a debugger on non-finite results (:178-182); here it is an error
        return out

    def per_channel_dynamic_calibration(self, x):
        with torch.no_grad():
            self.forward(x)

    def register_scales_and_zeros(self):
        self.register_buffer("scales", self.scale)
        self.register_buffer("zeros", self.zero_point)
        del self.scale
        del self.round_zero_point

    def free(self):
        self.cached_xmax = None
        self.cached_xmin = None
