"""Learnable equivalent transformations (LET): per-channel scale/shift migrations between a norm
and the linears that consume it, between v_proj and out_proj, and between q_proj and k_proj.
Restates models/transformation.py:5-114 of the reference; cheap elementwise torch ops on [C] vectors
and on the weight matrices (the weight fake-quant kernels then consume the transformed weights)."""
from __future__ import annotations

import torch


class _Truncate(torch.autograd.Function):
    """|x| < threshold -> sign(x)*threshold, identity gradient."""

    @staticmethod
    def forward(ctx, x, threshold):
        # torch.where instead of the reference's masked assignment: same values, no host synchronisation
        return torch.where(x.abs() < threshold, x.sign() * threshold, x)

    @staticmethod
    def backward(ctx, g):
        return g.clone(), None


def truncate_number(number, threshold=1e-2):
    """Keeps smoothing scales away from zero (threshold 1e-2, as the decoder layers of the reference see it)."""
    return _Truncate.apply(number, threshold)


def _has_bias(m):
    return hasattr(m, "bias") and m.bias is not None


def smooth_ln_fcs_temporary(ln, fcs, scales, shifts):
    """norm output (x - shift)/scale; consumers absorb W*scale and b + W@shift."""
    ln.use_temporary_parameter = True
    fcs = fcs if isinstance(fcs, list) else [fcs]
    ln.temp_bias = (ln.bias - shifts) / scales if _has_bias(ln) else (-1 * shifts) / scales
    ln.temp_weight = ln.weight / scales
    for fc in fcs:
        fc.use_temporary_parameter = True
        fc.temp_bias = fc.bias + fc.weight @ shifts if _has_bias(fc) else fc.weight @ shifts
        fc.temp_weight = fc.weight * scales.view(1, -1)


def smooth_fc_fc_temporary(fc1, fc2, scales, shifts=None):
    """fc1 rows are divided by scale (after subtracting shift from its bias), fc2 columns multiplied."""
    fc1.use_temporary_parameter = True
    fc2.use_temporary_parameter = True
    if hasattr(fc1, "temp_weight"):
        fc1.temp_bias = (fc1.temp_bias - shifts) / scales.view(-1)
        fc1.temp_weight = fc1.temp_weight / scales.view(-1, 1)
    else:
        fc1.temp_bias = fc1.bias / scales.view(-1)
        fc1.temp_weight = fc1.weight / scales.view(-1, 1)
    fc2.temp_bias = fc2.bias + fc2.weight @ shifts if _has_bias(fc2) else fc2.weight @ shifts
    fc2.temp_weight = fc2.weight * scales.view(1, -1)


def smooth_q_k_temporary(q_proj, k_proj, scales):
    q_proj.use_temporary_parameter = True
    k_proj.use_temporary_parameter = True
    q_proj.temp_weight = q_proj.temp_weight / scales.view(-1, 1)
    q_proj.temp_bias = q_proj.temp_bias / scales.view(-1)
    k_proj.temp_weight = k_proj.temp_weight * scales.view(-1, 1)
    k_proj.temp_bias = k_proj.temp_bias * scales.view(-1)


def smooth_ln_fcs_inplace(ln, fcs, scales, shifts):
    ln.use_temporary_parameter = False
    fcs = fcs if isinstance(fcs, list) else [fcs]
    if _has_bias(ln):
        ln.bias.sub_(shifts)
        ln.bias.div_(scales)
    else:
        del ln.bias
        ln.register_buffer("bias", (-1 * shifts) / scales)
    ln.weight.div_(scales)
    for fc in fcs:
        fc.use_temporary_parameter = False
        if _has_bias(fc):
            fc.bias.add_(fc.weight @ shifts)
        else:
            del fc.bias
            fc.register_buffer("bias", fc.weight @ shifts)
        fc.weight.mul_(scales.view(1, -1))


def smooth_fc_fc_inplace(fc1, fc2, scales, shifts=None):
    fc1.use_temporary_parameter = False
    fc2.use_temporary_parameter = False
    fc1.bias.sub_(shifts)
    fc1.bias.div_(scales.view(-1))
    fc1.weight.div_(scales.view(-1, 1))
    if _has_bias(fc2):
        fc2.bias.add_(fc2.weight @ shifts)
    else:
        del fc2.bias
        fc2.register_buffer("bias", fc2.weight @ shifts)
    fc2.weight.mul_(scales.view(1, -1))


def smooth_q_k_inplace(q_proj, k_proj, scales):
    q_proj.use_temporary_parameter = False
    k_proj.use_temporary_parameter = False
    q_proj.weight.div_(scales.view(-1, 1))
    q_proj.bias.div_(scales.view(-1))
    k_proj.weight.mul_(scales.view(-1, 1))
    k_proj.bias.mul_(scales.view(-1))
