"""OmniLayerNorm / OmniLlamaRMSNorm: drop-ins for the reference norms (quantize/omni_norm.py:11-71).

When a ``reorder_index`` buffer has been registered (R1/R4 reorder), the input channels are gathered
with the LDS-staged gather kernel before the normalisation, exactly where the reference calls
``torch.index_select`` on every forward."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as F


class OmniLayerNorm(nn.Module):
    def __init__(self, ori_layer_norm) -> None:
        super().__init__()
        self.use_act_quant = True
        self.register_buffer("weight", ori_layer_norm.weight)
        if ori_layer_norm.bias is not None:
            self.register_buffer("bias", ori_layer_norm.bias)
        else:
            self.bias = None
        self.eps = ori_layer_norm.eps
        self.norm_func = nn.functional.layer_norm
        self.normalized_shape = ori_layer_norm.normalized_shape
        self.use_temporary_parameter = False
        self._perm = None  # (source tensor id, int32 index, int32 inverse) cached on the device
        self.fused_gather = True  # gather + LayerNorm in one kernel when a reorder index is registered

    def _permutation(self, device):
        idx = self.reorder_index
        key = (idx.data_ptr(), idx._version, str(device))
        if self._perm is None or self._perm[0] != key:
            i32 = idx.to(device=device, dtype=torch.int32).contiguous()
            inv = torch.empty_like(i32)
            inv[idx.to(device).long()] = torch.arange(idx.numel(), dtype=torch.int32, device=device)
            self._perm = (key, i32, inv)
        return self._perm[1], self._perm[2]

    def forward(self, x):
        if self.use_temporary_parameter:
            weight = self.temp_weight
            bias = self.temp_bias
        else:
            weight = self.weight
            bias = self.bias
        if hasattr(self, "reorder_index") and x.ndim in (2, 3):
            i32, inv = self._permutation(x.device)
            if (self.fused_gather and tuple(self.normalized_shape) == (x.shape[-1],)
                    and F.gather_layernorm_supported(x, weight, bias)):
                # one pass: gather + normalisation.  Under autocast PyTorch runs layer_norm in fp32.
                amp = torch.is_autocast_enabled()
                out_dtype = torch.float32 if amp else (weight.dtype if weight is not None else x.dtype)
                if amp or weight is None or weight.dtype == x.dtype:
                    return F.GatherLayerNorm.apply(x, i32, inv, weight, bias, self.eps, out_dtype)
            x = F.GatherLastDim.apply(x, i32, inv)
        return self.norm_func(x, self.normalized_shape, weight, bias, eps=self.eps)

    def set_quant_state(self, use_weight_quant, use_act_quant):
        self.use_act_quant = use_act_quant


class OmniLlamaRMSNorm(nn.Module):
    fused = True  # one kernel per direction instead of the elementwise chain in forward (class-wide switch)

    def __init__(self, ori_norm, eps=1e-6):
        super().__init__()
        self.register_buffer("weight", ori_norm.weight)
        self.bias = None
        self.variance_epsilon = eps
        self.use_temporary_parameter = False

    def forward(self, hidden_states):
        if self.use_temporary_parameter:
            weight = self.temp_weight
            bias = self.temp_bias
        else:
            weight = self.weight
            bias = self.bias if hasattr(self, "bias") else None
        if self.fused and F.rmsnorm_supported(hidden_states, weight, bias):
            # every intermediate of the chain below is an fp32 tensor (type promotion), so fp16 vectors enter as fp32
            return F.RMSNorm.apply(hidden_states, weight.float(), None if bias is None else bias.float(),
                                   self.variance_epsilon)
        input_dtype = hidden_states.dtype
        variance = hidden_states.to(torch.float32).pow(2).mean(-1, keepdim=True)
        hidden_states = hidden_states * torch.rsqrt(variance + self.variance_epsilon)
        if bias is not None:
            return (weight * hidden_states + bias).to(input_dtype)
        return (weight * hidden_states).to(input_dtype)
