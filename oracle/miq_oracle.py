"""CPU oracle for the fake-quantization hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch restatement, in plain PyTorch CPU ops, of the arithmetic the reference
performs in its quantizer classes.  It exists so that the HIP kernels can be checked on the GPU box
(where the reference itself is not present).  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import it; the product package never does.

Pinning: every function here is compared bit-for-bit against golden vectors that were produced by
importing the reference's own modules on CPU (tests/golden/gen_golden.py, fixtures committed under
tests/golden/), see tests/test_oracle_golden.py.

The op ORDER matters: for float16 tensors PyTorch rounds to float16 after every elementwise op, so
each function below performs exactly the reference's sequence of tensor ops (division is a true
division, rounding is round-half-to-even) and lets torch do the per-op rounding.  Gradients are
obtained by autograd through that same sequence (straight-through rounding expressed with
``detach``), except for the exp-v2 quantizer whose reference backward is hand-written and is restated
here as a custom Function.

Reference locations (relative to the reference tree):
  uniform row-wise (LWC)            quantize/quantizer.py:99-121,248-273
  static activation calibration     quantize/quantizer.py:157-238
  exp/log v1                        quantize/quantizer_exp.py:92-166
  exp/log v2                        quantize/quantizer_exp_v2.py:21-105,186-204
  learnable mixed-bit               quantize/quantizer_mix.py:118-191,207-218
  fixed mixed-bit                   quantize/quantizer_mix_v2.py:91-146
  reorder index / stats / gather    quantize/reorder_utils.py:68-147, quantize/omni_norm.py:36-40
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np
import torch

CLIP_LO = 1e-5
CLIP_HI = 1e4


# --------------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------------
def _ste(fn, t: torch.Tensor) -> torch.Tensor:
    """value of fn(t), gradient of identity (quantize/quantizer.py:13-17)."""
    return (fn(t) - t).detach() + t


def _as_groups(x: torch.Tensor, group_size: Optional[int], pad_rows: bool = True):
    """View a [O, I] matrix as quantization rows.

    Without grouping the rows are the matrix rows.  With grouping every row is zero-padded on the
    right up to a multiple of ``group_size`` and cut into groups (quantizer.py:249-255); the padding
    takes part in the min/max.  The reference only computes the padding amount when LWC is on
    (quantizer.py:63-69): without LWC ``pad_rows`` is False and the matrix is cut into groups as one
    flat buffer, so groups may straddle rows.  Returns (rows2d, pad), ``pad`` = zero columns added.
    """
    if not group_size:
        return x, 0
    assert x.dim() == 2
    pad = (-x.shape[1]) % group_size if pad_rows else 0
    if pad:
        x = torch.cat((x, torch.zeros((x.shape[0], pad), dtype=x.dtype, device=x.device)), dim=1)
    return x.reshape(-1, group_size), pad


def _from_groups(y: torch.Tensor, shape, pad: int, group_size: Optional[int]):
    if not group_size:
        return y
    y = y.reshape(shape[0], shape[1] + pad)
    if pad:
        y = y[:, :-pad]
    return y


def _row_minmax_clipped(rows, up, low, activated=False):
    """Row extremes after learnable clipping (quantizer.py:256-261).  ``activated``: up / low already hold
    sigmoid(factor) -- used by tests that evaluate the sigmoid on the device under test, whose last bit may
    differ from the CPU's, and then need the rest of the op sequence for THOSE clip factors."""
    rmin = rows.amin(-1, keepdim=True)
    rmax = rows.amax(-1, keepdim=True)
    if up is not None:
        rmax = (up if activated else torch.sigmoid(up)) * rmax
        rmin = (low if activated else torch.sigmoid(low)) * rmin
    return rmin, rmax


def _affine_fq(rows, scale, rzp, qmax):
    """round(x/s) + zp -> clamp [0, qmax] -> - zp -> * s; returns (dequant, clamped codes)."""
    q = _ste(torch.round, rows / scale)
    q = q.add(rzp)
    if isinstance(qmax, torch.Tensor):
        qc = q.clamp(torch.zeros_like(qmax), qmax)
    else:
        qc = q.clamp(0, qmax)
    return qc.sub(rzp).mul(scale), qc


@dataclass
class RowResult:
    out: torch.Tensor
    scale: torch.Tensor
    rzp: Optional[torch.Tensor]
    codes: Optional[torch.Tensor] = None
    extra: dict = field(default_factory=dict)


# --------------------------------------------------------------------------------------------
# U1 + U2 (+ M2): uniform row-wise fake quant with learnable weight clipping
# --------------------------------------------------------------------------------------------
def uniform_row(x, n_bits, symmetric=False, up=None, low=None, group_size=None,
                clamp_asym_scale=False, activated=False) -> RowResult:
    """Row-wise ("per_channel"/"per_token" dynamic) uniform fake quantization.

    ``n_bits`` is an int, or a [rows, 1] tensor of per-row integer bit-widths (the fixed mixed-bit
    quantizer, quantizer_mix_v2.py:116-146, which also clamps the asymmetric scale:
    ``clamp_asym_scale=True``).  The plain uniform quantizer does NOT clamp the asymmetric scale
    (quantizer.py:270-271 overwrites the clamped value), so a constant row divides by zero.
    """
    rows, pad = _as_groups(x, group_size, up is not None)
    rmin, rmax = _row_minmax_clipped(rows, up, low, activated)
    if symmetric:
        amax = torch.max(rmax.abs(), rmin.abs())
        half = 2 ** (n_bits - 1) - 1
        scale = amax / half
        if not isinstance(n_bits, torch.Tensor):
            scale = scale.clamp(min=CLIP_LO, max=CLIP_HI)
        zp = half * torch.ones_like(scale)
    else:
        scale = (rmax - rmin) / (2 ** n_bits - 1)
        if clamp_asym_scale:
            scale = scale.clamp(min=CLIP_LO, max=CLIP_HI)
        zp = -(rmin) / scale
    rzp = zp.clamp(min=-CLIP_HI, max=CLIP_HI).round()
    qmax = 2 ** n_bits - 1
    deq, codes = _affine_fq(rows, scale, rzp, qmax)
    out = _from_groups(deq, x.shape, pad, group_size)
    codes = _from_groups(codes.detach(), x.shape, pad, group_size)
    return RowResult(out, scale, rzp, codes)


def fake_quant_given(x, scale, rzp, n_bits, group_size=None, deficiency=0):
    """UniformAffineQuantizer.fake_quant (quantizer.py:99-121) with parameters handed in: zero-pad every row by
    ``deficiency`` columns, view as groups, round(x / s) + zp -> clamp -> - zp -> * s, undo the view, strip the padding."""
    xr = x
    if deficiency > 0:
        xr = torch.cat((xr, torch.zeros((x.shape[0], deficiency), dtype=x.dtype)), dim=1)
    shape = xr.shape
    if group_size:
        assert xr.dim() == 2
        xr = xr.reshape(-1, group_size)
    deq, _codes = _affine_fq(xr, scale, rzp, 2 ** n_bits - 1)
    if group_size:
        deq = deq.reshape(shape)
    if deficiency > 0:
        deq = deq[:, :-deficiency]
    return deq


# --------------------------------------------------------------------------------------------
# E1: exp/log quantizer v1 (no zero level)
# --------------------------------------------------------------------------------------------
def exp1_row(x, n_bits, symmetric=True, up=None, low=None, group_size=None, activated=False) -> RowResult:
    rows, pad = _as_groups(x, group_size, up is not None)
    rmin, rmax = _row_minmax_clipped(rows, up, low, activated)
    if symmetric:
        scale = torch.max(rmax.abs(), rmin.abs()).clamp(min=CLIP_LO, max=CLIP_HI)
        mid = None
    else:
        scale = ((rmax - rmin) / 2).clamp(min=CLIP_LO, max=CLIP_HI)
        mid = (rmax + rmin) / 2
    e_lo = -(2 ** (n_bits - 1) - 1)
    v = rows if mid is None else rows - mid
    sgn = torch.sign(v)
    mag = torch.abs(v / scale).clamp(CLIP_LO, CLIP_HI)
    e = _ste(torch.round, torch.log2(mag)).clamp(e_lo, 0)
    deq = torch.pow(2.0, e) * sgn * scale
    if mid is not None:
        deq = deq + mid
    out = _from_groups(deq, x.shape, pad, group_size)
    return RowResult(out, scale, mid, _from_groups(e.detach(), x.shape, pad, group_size),
                     {"sign": _from_groups(sgn.detach(), x.shape, pad, group_size)})


# --------------------------------------------------------------------------------------------
# E2: exp/log quantizer v2 (zero level, hand-written backward)
# --------------------------------------------------------------------------------------------
def exp2_levels(n_bits: int):
    top = 1 << ((1 << (n_bits - 1)) - 2)
    return -top, top


class _Exp2FQ(torch.autograd.Function):
    """sign * clamp(2^round(log2(|q|/1.5)+0.5), ±top) * s with q = x/s; |q| < 0.5 -> 0.

    The backward is the reference's own formula (quantizer_exp_v2.py:65-105), not the true
    derivative: d/dq is (level / |q|) where the level is un-clamped, non-zero and above threshold.
    """

    @staticmethod
    def forward(ctx, x, scale, n_bits):
        lo, hi = exp2_levels(n_bits)
        q = x / scale
        sgn = q.sign()
        mag = q.abs()
        is_zero = mag < 0.5
        mag = mag.clamp(1e-4, 1e4)
        t = torch.log2(mag / 1.5) + 0.5
        below = t < -0.5
        t = torch.where(below, torch.zeros_like(t), t)
        lvl = 2 ** t.round()
        lvl = torch.where(is_zero, torch.zeros_like(lvl), lvl)
        signed = lvl.clamp(lo, hi) * sgn
        ctx.save_for_backward(x, scale, sgn, mag, lvl, signed, is_zero, below)
        ctx.hi = hi
        ctx.lo = lo
        return signed * scale

    @staticmethod
    def backward(ctx, g):
        x, scale, sgn, mag, lvl, signed, is_zero, below = ctx.saved_tensors
        g = g.expand(x.shape)
        gs = g * scale
        d_scale_a = g * signed
        gq = torch.where(sgn == -1, gs * -1, gs)
        dead = lvl.ge(ctx.hi + 0.0001) | lvl.le(ctx.lo + 0.0001) | is_zero
        gq = torch.where(dead, torch.zeros_like(gq), gq)
        gt = math.log(2) * lvl * gq
        gt = torch.where(below, torch.zeros_like(gt), gt)
        gmag = gt / math.log(2) / mag
        gqq = gmag * sgn
        return gqq / scale, d_scale_a + gqq * (-x / scale ** 2), None


def exp2_row(x, n_bits, up=None, low=None, group_size=None, activated=False) -> RowResult:
    """Group handling mirrors the reference: calibration sees the grouped view but the fake-quant
    itself is applied to the un-grouped tensor with the [rows,1] scale broadcast
    (quantizer_exp_v2.py:174-176,186-204), so it is only shape-consistent when group_size is None.
    """
    rows, _pad = _as_groups(x, group_size, up is not None)
    rmin, rmax = _row_minmax_clipped(rows, up, low, activated)
    lo, hi = exp2_levels(n_bits)
    scale = (torch.max(rmax.abs(), rmin.abs()) / (hi - lo)).clamp(min=CLIP_LO, max=CLIP_HI)
    out = _Exp2FQ.apply(x, scale, n_bits)
    with torch.no_grad():
        lv = (out / scale).detach()
    return RowResult(out, scale, None, lv)


# --------------------------------------------------------------------------------------------
# M1: learnable per-row bit-width (FracBits blend of floor/ceil bit-widths)
# --------------------------------------------------------------------------------------------
def mix_row(x, n_bits_param, symmetric=False, up=None, low=None, group_size=None,
            hard=False, threshold=0.0, activated=False) -> RowResult:
    """``n_bits_param`` [rows,1] float; the caller is expected to have clamped it to [2, 8]
    (the reference clamps ``.data`` in place at the top of forward, quantizer_mix.py:145)."""
    n = n_bits_param
    if hard:
        n = _ste(torch.round, n + threshold)
    rows, pad = _as_groups(x, group_size, up is not None)
    rmin, rmax = _row_minmax_clipped(rows, up, low, activated)
    n_lo = _ste(torch.floor, n)
    n_hi = _ste(torch.ceil, n)
    q_lo = 2 ** n_lo - 1
    q_hi = 2 ** n_hi - 1
    w_hi = n - n_lo
    w_lo = 1.0 - w_hi
    if symmetric:
        amax = torch.max(rmax.abs(), rmin.abs())
        s_lo = amax / (2 ** (n_lo - 1) - 1)
        s_hi = amax / (2 ** (n_hi - 1) - 1)
        z_lo = (2 ** (n_lo - 1) - 1) * torch.ones_like(s_lo)
        z_hi = (2 ** (n_hi - 1) - 1) * torch.ones_like(s_hi)
    else:
        rng = rmax - rmin
        s_lo = (rng / (2 ** n_lo - 1)).clamp(min=CLIP_LO, max=CLIP_HI)
        s_hi = (rng / (2 ** n_hi - 1)).clamp(min=CLIP_LO, max=CLIP_HI)
        z_lo = -rmin / s_lo
        z_hi = -rmin / s_hi
    rz_lo = z_lo.clamp(min=-CLIP_HI, max=CLIP_HI).round()
    rz_hi = z_hi.clamp(min=-CLIP_HI, max=CLIP_HI).round()
    d_lo, c_lo = _affine_fq(rows, s_lo, rz_lo, q_lo)
    d_hi, c_hi = _affine_fq(rows, s_hi, rz_hi, q_hi)
    d_lo = _from_groups(d_lo, x.shape, pad, group_size)
    d_hi = _from_groups(d_hi, x.shape, pad, group_size)
    out = w_lo * d_lo + w_hi * d_hi
    return RowResult(out, s_lo, rz_lo, _from_groups(c_lo.detach(), x.shape, pad, group_size),
                     {"scale_hi": s_hi, "rzp_hi": rz_hi,
                      "codes_hi": _from_groups(c_hi.detach(), x.shape, pad, group_size)})


def mix_model_size(n_bits_param, shape, target_bit, hard=False, threshold=0.0):
    """(model_size, target_size) scalars: mean bits per weight and the target (quantizer_mix.py:207-218)."""
    n = n_bits_param
    if hard:
        n = _ste(torch.round, n + threshold)
    prop = shape[0] * shape[1]
    target = torch.tensor(shape[0] * shape[1] * target_bit) / prop
    size = torch.sum(n * shape[1]) / prop
    return size, target


# --------------------------------------------------------------------------------------------
# U4 + U5: static activation quantisation (per tensor / per cluster, minmax or EMA)
# --------------------------------------------------------------------------------------------
@dataclass
class ActState:
    """What the reference keeps on the quantizer object between calls."""
    cached_min: Optional[torch.Tensor] = None
    cached_max: Optional[torch.Tensor] = None
    scale: Optional[torch.Tensor] = None
    zero_point: Optional[torch.Tensor] = None
    rzp: Optional[torch.Tensor] = None


def act_calibrate(x, state: ActState, n_bits, method="per_cluster", metric="minmax",
                  cluster_dim=None, cluster_counts: Optional[Sequence[int]] = None):
    """Update ``state`` from one batch (quantizer.py:157-238)."""
    if method == "per_tensor":
        cmin = torch.min(x)
        cmax = torch.max(x)
    else:
        axes = [a for a in range(x.ndim) if a != cluster_dim]
        cmin = x.amin(axes, keepdim=True).clone()
        cmax = x.amax(axes, keepdim=True).clone()
    if state.cached_max is not None:
        if metric == "minmax":
            cmax = torch.max(state.cached_max, cmax)
            cmin = torch.min(state.cached_min, cmin)
        if metric == "ema_minmax":
            cmax = state.cached_max * 0.99 + cmax * 0.01
            cmin = state.cached_min * 0.99 + cmin * 0.01
    state.cached_max = cmax
    state.cached_min = cmin
    if method != "per_tensor" and cluster_dim is not None:
        # the cached tensors alias cmin/cmax and are overwritten in place, exactly like the
        # reference's narrow().copy_() (quantizer.py:203-210)
        start = 0
        for cnt in cluster_counts:
            cnt = int(cnt)
            seg_min = torch.narrow(cmin, cluster_dim, start, cnt).clone().amin(cluster_dim, keepdim=True)
            seg_max = torch.narrow(cmax, cluster_dim, start, cnt).clone().amax(cluster_dim, keepdim=True)
            cmin.narrow(cluster_dim, start, cnt).copy_(seg_min)
            cmax.narrow(cluster_dim, start, cnt).copy_(seg_max)
            start += cnt
    scale = (cmax - cmin) / (2 ** n_bits - 1)
    scale.clamp_(min=CLIP_LO, max=CLIP_HI)
    zp = -(cmin) / scale
    zp.clamp_(min=-CLIP_HI, max=CLIP_HI)
    state.scale = scale
    state.zero_point = zp
    state.rzp = zp.round()
    return state


def act_quant(x, state: ActState, n_bits):
    deq, codes = _affine_fq(x, state.scale, state.rzp, 2 ** n_bits - 1)
    return deq, codes.detach()


def act_forward(x, state: ActState, n_bits, mode="calibration", **kw):
    """Calibration mode updates the state AND quantizes (quantizer.py:134-140)."""
    if mode == "calibration":
        act_calibrate(x, state, n_bits, **kw)
    return act_quant(x, state, n_bits)


# --------------------------------------------------------------------------------------------
# R1-R3: reorder statistics, index and gather
# --------------------------------------------------------------------------------------------
def channel_minmax(t: torch.Tensor):
    """Per-last-dim-channel (max, min) over all leading axes (reorder_utils.py:68-109)."""
    axes = list(range(t.ndim - 1))
    return torch.amax(t, axes), torch.amin(t, axes)


def running_minmax(prev, new):
    if prev is None:
        return new
    return torch.max(prev[0], new[0]), torch.min(prev[1], new[1])


def reorder_index(xmax, xmin, n_clusters: int, n_heads: Optional[int] = None, stable=True):
    """Ascending argsort of the per-channel range inside each head; equal-size clusters.

    With a list input only the first entry decides the order (reorder_utils.py:119-125).
    ``stable=True`` is this build's documented tie rule (ascending, ties by channel id); the
    reference calls torch.argsort without ``stable`` so ties are unspecified there.
    """
    n_heads = n_heads or 1
    if isinstance(xmax, (list, tuple)):
        xmax, xmin = xmax[0], xmin[0]
    rng = (xmax - xmin).reshape(n_heads, -1)
    per = rng.shape[1]
    idx = []
    for h in range(n_heads):
        idx.append(torch.argsort(rng[h], stable=stable) + h * per)
    counts = np.tile(np.array([per // n_clusters] * n_clusters), n_heads)
    return torch.hstack(idx), counts


def gather_lastdim(x, index):
    return torch.index_select(x, x.ndim - 1, index)


# --------------------------------------------------------------------------------------------
# activation scale / shift statistics and sensitivity-driven bit allocation
# --------------------------------------------------------------------------------------------
def act_scale_update(prev: Optional[torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """Running per-channel abs-max of a Linear input (generate_act_scale_shift.py:31-38)."""
    t = x.reshape(-1, x.shape[-1]).abs().detach()
    coming = torch.max(t, dim=0)[0].float().cpu()
    return coming if prev is None else torch.max(prev, coming)


def act_shift_update(prev: Optional[torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """0.99/0.01 EMA of the per-channel (max+min)/2 of a Linear input (generate_act_scale_shift.py:64-72)."""
    t = x.reshape(-1, x.shape[-1]).detach()
    mx = torch.max(t, dim=0)[0].float().cpu()
    mn = torch.min(t, dim=0)[0].float().cpu()
    mid = (mx + mn) / 2
    return mid if prev is None else 0.99 * prev + 0.01 * mid


def row_absmean(g: torch.Tensor) -> torch.Tensor:
    """Per-output-channel sensitivity term of one sample (get_grads.py:204)."""
    return torch.mean(torch.abs(g), dim=1)


def allocate_bits(sensitivity: torch.Tensor) -> torch.Tensor:
    """10 % most sensitive channels 4 bits, 10 % least sensitive 2 bits, the rest 3 (get_grads.py:222-230).
    Parity unpinned: the reference holds this inside a script body that cannot be imported."""
    _v, indices = torch.sort(sensitivity, descending=True)
    n = sensitivity.shape[0]
    bits = torch.zeros_like(sensitivity)
    bits[indices[0:int(n * 0.1)]] = 4
    bits[indices[int(n * 0.1):int(n * 0.9)]] = 3
    bits[indices[int(n * 0.9):]] = 2
    return bits
