"""QuantLinear: drop-in for the reference wrapper (quantize/int_linear.py:14-103).

Holds the original weight/bias as buffers, owns a weight quantizer (uniform / exponential / learnable
mixed-bit / fixed mixed-bit) and an activation quantizer, and runs ``F.linear`` on whatever
combination of (temporary | quantized | plain) weight and (quantized | plain) input is selected."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as TF

from . import functional as F
from .quantizer import UniformAffineQuantizer
from .quantizer_exp_v2 import ExpAffineQuantizer
from .quantizer_mix import MixAffineQuantizer
from .quantizer_mix_v2 import Mix2AffineQuantizer


class QuantLinear(nn.Module):
    fused_bias_grad = True  # bias gradient by the column-sum kernel when the bias is being trained (class-wide switch)

    def __init__(
        self,
        org_module: nn.Linear,
        weight_quant_params: dict = {},
        act_quant_params: dict = {},
        disable_input_quant=False,
        weight_exp_quant=False,
        weight_mix_quant=False,
        weight_mix2_quant=False,
        weight_nbits: torch.Tensor = [],
    ):
        super().__init__()
        self.fwd_kwargs = dict()
        self.fwd_func = TF.linear
        # plain tensors sharing the original storage: a buffer that is still an nn.Parameter would turn
        # `temp_weight = weight` into a parameter registration (the reference avoids that by casting
        # the block .half() then .float(), which strips the Parameter type as a side effect)
        self.register_buffer("weight", org_module.weight.detach())
        if org_module.bias is not None:
            self.register_buffer("bias", org_module.bias.detach())
        else:
            self.bias = None
        self.in_features = org_module.in_features
        self.out_features = org_module.out_features
        self.use_weight_quant = False
        self.use_act_quant = False
        shape = org_module.weight.shape
        if weight_exp_quant:
            self.weight_quantizer = ExpAffineQuantizer(**weight_quant_params, shape=shape)
        elif weight_mix_quant:
            self.weight_quantizer = MixAffineQuantizer(**weight_quant_params, shape=shape)
        elif weight_mix2_quant:
            self.weight_quantizer = Mix2AffineQuantizer(**weight_quant_params, shape=shape, mix2_nbits=weight_nbits)
        else:
            self.weight_quantizer = UniformAffineQuantizer(**weight_quant_params, shape=shape)
        if not disable_input_quant:
            self.act_quantizer = UniformAffineQuantizer(**act_quant_params)
        else:
            self.act_quantizer = None
        self.disable_input_quant = disable_input_quant
        self.use_temporary_parameter = False

    def forward(self, input: torch.Tensor, input_quantized: bool = False):
        """``input_quantized``: the caller already applied this layer's activation quantizer (one evaluation shared by
        the layers that read the same tensor, see quantizer.shared_static_quant)."""
        if self.use_temporary_parameter:
            weight = self.temp_weight
            bias = self.temp_bias
        elif self.use_weight_quant:
            weight = self.weight_quantizer(self.weight)
            bias = self.bias
        else:
            weight = self.weight
            bias = self.bias
        if self.use_act_quant and not self.disable_input_quant and not input_quantized:
            input = self.act_quantizer(input)
        if (self.fused_bias_grad and bias is not None and bias.requires_grad and torch.is_grad_enabled()
                and self.fwd_func is TF.linear and not self.fwd_kwargs and F.linear_supported(input, weight, bias)):
            return F.LinearBias.apply(input, weight, bias)
        return self.fwd_func(input, weight, bias, **self.fwd_kwargs)

    def set_ic_cluster_counts(self, counts, w_dim=1, a_dim=2):
        self.weight_quantizer.cluster_dim = w_dim       # recorded only: weights are quantized per output row
        self.weight_quantizer.cluster_counts = counts
        if a_dim is not None:
            self.act_quantizer.cluster_dim = a_dim
            self.act_quantizer.cluster_counts = counts

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self.use_weight_quant = weight_quant
        self.use_act_quant = act_quant
