"""Shared test helpers: golden fixture loading, input regeneration, ulp distance."""
from __future__ import annotations

import functools
import os

import numpy as np
import torch

from miq import synth

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
DT = {"f32": torch.float32, "f16": torch.float16}


@functools.lru_cache(maxsize=None)
def golden(family: str):
    return torch.load(os.path.join(GOLDEN, f"{family}.pt"), weights_only=True)


def n_groups(shape, group):
    if not group:
        return shape[0]
    return shape[0] * int(np.ceil(shape[1] / group))


def weight_inputs(meta, groups=None):
    """Re-create (x, up, low, w) of a golden weight case from its meta (same recipe as the generator)."""
    shape, dtype, seed = tuple(meta["shape"]), DT[meta["dtype"]], meta["seed"]
    if meta.get("edge"):
        x = synth.weight_with_edge_rows(shape[0], shape[1], seed, dtype)
    else:
        x = synth.weight(shape[0], shape[1], seed, dtype)
    up = low = None
    if meta.get("lwc"):
        g = groups if groups is not None else n_groups(shape, meta.get("group_size", 0))
        up, low = synth.lwc_factors(g, seed + 1, dtype)
    w = synth.grad_like(x, seed + 2)
    return x, up, low, w


def gpu_sigmoid(*ts):
    """torch.sigmoid evaluated on the GPU (what a module running there computes, and what the reference would), back on the
    CPU: the clip factors the oracle needs to restate the module's arithmetic for."""
    return tuple(None if t is None else torch.sigmoid(t.to("cuda:0")).cpu() for t in ts)


def same(a: torch.Tensor, b: torch.Tensor) -> bool:
    """Bitwise equality of values, NaNs compare equal (payload/sign of NaN ignored)."""
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    an, bn = torch.isnan(a), torch.isnan(b)
    if not torch.equal(an, bn):
        return False
    return torch.equal(torch.where(an, torch.zeros_like(a), a), torch.where(bn, torch.zeros_like(b), b))


def same_act_backward(dx, ref_dx, x, cmin, cmax, why=""):
    """Two forms of the calibrating activation quantizer's backward that sum the scale gradient in different orders
    (per-lane partials folded per cluster against per-channel partials folded per channel): every element that is NOT
    at an extreme of its cluster carries the direct path only and must be bit-identical; the elements at the extremes
    (the only ones a share of the scale gradient is routed to) must agree to the rounding of the sum -- which also pins
    the tie counts, since a share divided by a wrong count is off by a factor."""
    dx, ref_dx, x = dx.detach().cpu(), ref_dx.detach().cpu(), x.detach().cpu()
    cmin, cmax = cmin.detach().cpu().reshape(1, -1), cmax.detach().cpu().reshape(1, -1)
    tied = (x == cmax) | (x == cmin)
    assert same(torch.where(tied, torch.zeros_like(dx), dx), torch.where(tied, torch.zeros_like(ref_dx), ref_dx)), \
        f"{why}: direct path differs away from the cluster extremes"
    if bool(tied.any()):
        a, b = dx[tied].double(), ref_dx[tied].double()
        ok = torch.isfinite(b)
        assert torch.equal(torch.isfinite(a), ok), f"{why}: finiteness of the routed elements differs"
        if bool(ok.any()):
            # norm-wise, like every gradient bound of these tests (dS is a sum with cancellation over the whole cluster
            # slab; against the oracle's autograd the routed elements are held to 5e-3)
            tol = 1e-4 if dx.dtype == torch.float32 else 2e-3
            err = ((a[ok] - b[ok]).abs().max() / b[ok].abs().max().clamp_min(1e-30)).item()
            assert err <= tol, f"{why}: routed elements differ by {err:.3g} (tolerance {tol})"


def ulp_diff(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Distance in units-in-the-last-place between same-dtype float tensors (NaN==NaN -> 0, sign-aware)."""
    assert a.dtype == b.dtype
    a, b = a.detach().cpu().contiguous(), b.detach().cpu().contiguous()
    if a.dtype == torch.float32:
        ia, ib = a.view(torch.int32).to(torch.int64), b.view(torch.int32).to(torch.int64)
        sign = 1 << 31
    else:
        ia, ib = a.view(torch.int16).to(torch.int64), b.view(torch.int16).to(torch.int64)
        sign = 1 << 15
    # map sign-magnitude to a monotone integer line
    ia = torch.where(ia < 0, -(ia + sign), ia)
    ib = torch.where(ib < 0, -(ib + sign), ib)
    d = (ia - ib).abs()
    both_nan = torch.isnan(a) & torch.isnan(b)
    one_nan = torch.isnan(a) ^ torch.isnan(b)
    d = torch.where(both_nan, torch.zeros_like(d), d)
    d = torch.where(one_nan, torch.full_like(d, 1 << 40), d)
    return d


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a-b| / max |b| in float64 (norm-wise; robust for sums with cancellation)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    a, b = torch.nan_to_num(a, nan=0.0, posinf=0.0, neginf=0.0), torch.nan_to_num(b, nan=0.0, posinf=0.0, neginf=0.0)
    den = b.abs().max().item()
    return (a - b).abs().max().item() / (den if den > 0 else 1.0)


class StatToy(torch.nn.Module):
    """Toy model for the activation-statistics fixtures: every nn.Linear input is an exact elementwise
    function of the sample, so the hooked tensors are the same bits on every device."""
    def __init__(self, c=96, k=40, seed=7):
        super().__init__()
        g = synth.gen(seed)
        self.a = torch.nn.Linear(c, 8)
        self.b = torch.nn.Linear(c, 8)
        self.c = torch.nn.Linear(k, 8, bias=False)
        self.k = k
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)

    def forward(self, x):
        return self.a(x).sum() + self.b(torch.relu(x) * 0.5 + 1.0).sum() + self.c(-x[..., :self.k]).sum()
