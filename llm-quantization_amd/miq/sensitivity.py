"""Gradient sensitivity -> per-output-channel bit allocation for the fixed mixed-bit quantizer.

Restates the per-layer body of the reference's get_grads.py:170-231: with the weights of a block
made trainable, the block's MSE loss (x10) against its own full-precision output is back-propagated
for every calibration sample and ``mean(|dW|, dim=1)`` is accumulated per output channel; the
concatenated sensitivities are sorted and the top 10 % of the channels get 4 bits, the bottom 10 %
2 bits, the rest 3.  The result has the layout of ``grads_results_uniform/layer{N}_nbits.pt`` that
the reference's decoder layers slice by fixed offsets (models/int_opt_layer.py:46-75, 275-300):
[fc1 | fc2 | q_proj | k_proj | v_proj | out_proj].
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch
import torch.nn as nn

from . import functional as F
from .int_linear import QuantLinear
from .int_matmul import QuantMatMul

OPT_ORDER = ("fc1", "fc2", "self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.out_proj")
LLAMA_ORDER = ("mlp.gate_proj", "mlp.up_proj", "mlp.down_proj", "self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj",
               "self_attn.o_proj")


def allocate_bits(sensitivity: torch.Tensor, high_frac: float = 0.1, low_frac: float = 0.1, bits=(4, 3, 2)) -> torch.Tensor:
    """get_grads.py:222-230: sort descending; first ``high_frac`` -> bits[0], last ``low_frac`` ->
    bits[2], the rest bits[1].  Returns a float tensor like the reference's file."""
    _sorted, indices = torch.sort(sensitivity, descending=True)
    n = sensitivity.shape[0]
    n_bits = torch.zeros_like(sensitivity)
    n_bits[indices[0:int(n * high_frac)]] = bits[0]
    n_bits[indices[int(n * high_frac):int(n * (1 - low_frac))]] = bits[1]
    n_bits[indices[int(n * (1 - low_frac)):]] = bits[2]
    return n_bits


def split_bits(n_bits: torch.Tensor, sizes: Sequence[int], names: Sequence[str]) -> Dict[str, torch.Tensor]:
    out, start = {}, 0
    for name, size in zip(names, sizes):
        out[name] = n_bits[start:start + size]
        start += size
    return out


def _linears(qlayer, order) -> List[QuantLinear]:
    mods = dict(qlayer.named_modules())
    return [mods[n] for n in order]


def layer_sensitivity(qlayer, inps, attention_mask=None, position_ids=None, order=OPT_ORDER, batch_size: int = 1,
                      loss_scale: float = 10.0):
    """Accumulated mean |dW| per output channel for the linears named in ``order``.

    ``inps``: [N, T, H] block inputs (fp32).  The block must be in fp32 with activation quantization
    on and weight quantization off (get_grads.py:186-187); its weights are turned into Parameters for
    the duration of the call.  Returns (list of fp32 [out_features] tensors, block outputs [N, T, H]).
    """
    mods = _linears(qlayer, order)
    dev = inps.device
    n = inps.shape[0]
    qlayer.set_quant_state(weight_quant=False, act_quant=False)
    qlayer.eval()
    targets = torch.empty_like(inps)
    with torch.no_grad():
        for j in range(n):
            targets[j] = qlayer(inps[j].unsqueeze(0), attention_mask=attention_mask, position_ids=position_ids)[0]
    qlayer.train()
    with torch.no_grad():
        qlayer.float()
    qlayer.set_quant_state(weight_quant=False, act_quant=True)
    saved = []
    for m in mods:  # buffers -> Parameters so that autograd produces dW
        w = m.weight
        del m._buffers["weight"]
        m.weight = nn.Parameter(w.detach().float())
        saved.append(w)
    acc = [torch.zeros(m.weight.shape[0], dtype=torch.float32, device=dev) for m in mods]
    mask_b = attention_mask.repeat(batch_size, 1, 1, 1) if attention_mask is not None else None
    try:
        for j in range(n // batch_size):
            i0 = j * batch_size
            out = qlayer(inps[i0:i0 + batch_size], attention_mask=mask_b, position_ids=position_ids)[0]
            loss = torch.nn.functional.mse_loss(targets[i0:i0 + batch_size], out) * loss_scale
            loss.backward()
            for m, a in zip(mods, acc):
                F.row_absmean_accum(m.weight.grad.contiguous(), a)
                m.weight.grad = None
            for _n, q in qlayer.named_modules():
                if isinstance(q, QuantLinear) and not q.disable_input_quant:
                    q.act_quantizer.free()
                if isinstance(q, QuantMatMul):
                    q.x1_quantizer.free()
                    q.x2_quantizer.free()
    finally:
        for m, w in zip(mods, saved):
            del m._parameters["weight"]
            m.register_buffer("weight", w)
    return acc, targets


def layer_bit_allocation(qlayer, inps, attention_mask=None, position_ids=None, order=OPT_ORDER, **kw):
    """The tensor the reference stores as ``layer{N}_nbits.pt`` plus the per-linear split."""
    acc, targets = layer_sensitivity(qlayer, inps, attention_mask, position_ids, order, **kw)
    n_bits = allocate_bits(torch.cat(acc, dim=0))
    return n_bits, split_bits(n_bits, [a.numel() for a in acc], order), targets
