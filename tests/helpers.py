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
