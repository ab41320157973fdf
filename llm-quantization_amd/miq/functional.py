"""Tensor-level entry points: thin wrappers that validate shapes, allocate outputs with torch and
call the C ABI, plus the autograd Functions that pair each forward kernel with its backward kernel.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib as L


def _rows_of(x: torch.Tensor, group: int, padded: bool):
    """(O, I, group_arg, n_rows) for a row-wise call.

    ``padded`` mirrors the reference: only quantizers built with LWC know the padding amount
    (quantize/quantizer.py:63-69); otherwise the buffer is cut into groups flat.
    """
    I = x.shape[-1]
    O = x.numel() // I if I else 0
    if not group:
        return O, I, 0, O
    if padded:
        G = -(-I // group)
        return O, I, group, O * G
    if x.numel() % group:
        raise L.MiqError(f"tensor of {x.numel()} elements cannot be cut into groups of {group}")
    return x.numel() // group, group, 0, x.numel() // group


def _flags(clamp_asym: bool, preactivated: bool) -> int:
    return (1 if clamp_asym else 0) | (2 if preactivated else 0)


def rowquant_uniform_fwd(x, n_bits: int, symmetric: bool, up=None, low=None, group: int = 0,
                         row_bits: Optional[torch.Tensor] = None, clamp_asym: bool = False,
                         want_codes: bool = False, preactivated: bool = False):
    """Returns (out, scale[rows,1], rzp[rows,1], codes|None).

    ``preactivated``: up/low already hold sigmoid(factor) (see MIQ_LWC_PREACTIVATED in miq.h).
    """
    dev = L.check_gpu(x, up, low, row_bits)
    O, I, g, rows = _rows_of(x, group, up is not None)
    out = torch.empty_like(x)
    scale = torch.empty((rows, 1), dtype=x.dtype, device=dev)
    rzp = torch.empty((rows, 1), dtype=x.dtype, device=dev)
    codes = torch.empty(x.shape, dtype=torch.uint8, device=dev) if want_codes else None
    if up is not None and (up.numel() != rows or low.numel() != rows or up.dtype != x.dtype or low.dtype != x.dtype):
        raise L.MiqError(f"LWC factors must be {rows} values of dtype {x.dtype}")
    if row_bits is not None and (row_bits.numel() != rows or row_bits.dtype != torch.float32):
        raise L.MiqError(f"row_bits must be {rows} float32 values")
    with torch.cuda.device(dev):
        L.call("miq_rowquant_uniform_fwd", L.ptr(x), L.ptr(up), L.ptr(low), L.ptr(row_bits), L.ptr(out),
               L.ptr(scale), L.ptr(rzp), L.ptr(codes), L.i64(O), L.i64(I), L.i64(g), L.i32(int(n_bits)),
               L.i32(int(bool(symmetric))), L.i32(_flags(clamp_asym, preactivated)), L.i32(L.dtype_code(x)),
               L.stream_of(x))
    return out, scale, rzp, codes


def rowquant_uniform_bwd(grad_out, x, n_bits: int, symmetric: bool, up=None, low=None, group: int = 0,
                         row_bits=None, clamp_asym: bool = False, want_dx: bool = True,
                         preactivated: bool = False):
    """Returns (dx|None, d_up|None, d_low|None)."""
    dev = L.check_gpu(grad_out, x, up, low, row_bits)
    O, I, g, rows = _rows_of(x, group, up is not None)
    dx = torch.empty_like(x) if want_dx else None
    d_up = torch.empty_like(up) if up is not None else None
    d_low = torch.empty_like(low) if low is not None else None
    with torch.cuda.device(dev):
        L.call("miq_rowquant_uniform_bwd", L.ptr(grad_out), L.ptr(x), L.ptr(up), L.ptr(low), L.ptr(row_bits),
               L.ptr(dx), L.ptr(d_up), L.ptr(d_low), L.i64(O), L.i64(I), L.i64(g), L.i32(int(n_bits)),
               L.i32(int(bool(symmetric))), L.i32(_flags(clamp_asym, preactivated)), L.i32(L.dtype_code(x)),
               L.stream_of(x))
    return dx, d_up, d_low


class UniformRowFQ(torch.autograd.Function):
    """out, scale, rzp = fq(x, up, low); gradients for x, up, low."""

    @staticmethod
    def forward(ctx, x, up, low, n_bits, symmetric, group, row_bits, clamp_asym):
        x = x.contiguous()
        out, scale, rzp, _ = rowquant_uniform_fwd(x, n_bits, symmetric, up, low, group, row_bits, clamp_asym)
        ctx.save_for_backward(x, up, low, row_bits)
        ctx.cfg = (n_bits, symmetric, group, clamp_asym)
        ctx.mark_non_differentiable(scale, rzp)
        return out, scale, rzp

    @staticmethod
    def backward(ctx, g, _gs, _gz):
        x, up, low, row_bits = ctx.saved_tensors
        n_bits, symmetric, group, clamp_asym = ctx.cfg
        need_dx = ctx.needs_input_grad[0]
        need_p = up is not None and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        if not (need_dx or need_p):
            return (None,) * 8
        dx, d_up, d_low = rowquant_uniform_bwd(g.contiguous(), x, n_bits, symmetric, up, low, group, row_bits,
                                               clamp_asym, want_dx=need_dx)
        return dx, d_up, d_low, None, None, None, None, None
