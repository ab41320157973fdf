"""QuantMatMul: drop-in for the reference wrapper (quantize/int_matmul.py:7-59): two activation
quantizers for the operands of a batched matmul (q·kᵀ and p·v in attention)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .quantizer import UniformAffineQuantizer


class QuantMatMul(nn.Module):
    def __init__(
        self,
        x1_quant_params: dict = {},
        x2_quant_params: dict = {},
        disable_act_quant=False,
        matmul_func=torch.bmm,
    ):
        super().__init__()
        self.use_act_quant = False
        self.i_cluster_counts = None
        self.x1_quantizer = UniformAffineQuantizer(**x1_quant_params)
        self.x2_quantizer = UniformAffineQuantizer(**x2_quant_params)
        self.matmul_func = matmul_func
        self.disable_act_quant = disable_act_quant

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self.use_weight_quant = weight_quant
        self.use_act_quant = act_quant

    def quant_x1(self, x1):
        if self.use_act_quant:
            x1 = self.x1_quantizer(x1)
        return x1

    def quant_x2(self, x2):
        if self.use_act_quant:
            x2 = self.x2_quantizer(x2)
        return x2

    def forward(self, x1, x2):
        return self.matmul_func(x1, x2)

    def set_ic_cluster_counts(self, counts, x1_dim=2, x2_dim=2, cluster_x1=True, cluster_x2=True):
        if cluster_x1:
            self.x1_quantizer.cluster_dim = x1_dim
            self.x1_quantizer.cluster_counts = counts
        if cluster_x2:
            self.x2_quantizer.cluster_dim = x2_dim
            self.x2_quantizer.cluster_counts = counts
