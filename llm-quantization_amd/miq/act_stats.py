"""Per-input-channel activation statistics of every nn.Linear in a model: the abs-max that seeds the
LET scales and the EMA of the (max+min)/2 midpoint that seeds the LET shifts.

Drop-in for the reference's generate_act_scale_shift.py:26-95 (same function names, same result
dictionaries ``{module name: float32 CPU tensor [in_features]}``, the format of ``act_scales/*.pt``
and ``act_shifts/*.pt``).  Differences in mechanism, not in values:

* one streaming pass of the column-statistics kernel per hooked input yields min and max together
  (abs-max over rows == max(max, -min) exactly), so scales and shifts can share ONE model pass
  (``get_act_scales_and_shifts``) instead of two;
* the running statistics stay on the GPU and are copied to the host once at the end, instead of one
  blocking ``.cpu()`` per module per sample.
"""
from __future__ import annotations

import functools

import torch
import torch.nn as nn

from . import functional as F


def _stat_input(store, want_scale, want_shift, m, x, y, name):
    if isinstance(x, tuple):
        x = x[0]
    t = x.detach().reshape(-1, x.shape[-1])
    if t.dtype not in (torch.float32, torch.float16):
        t = t.float()
    mn, mx = F.channel_minmax(t.contiguous())
    # the reference takes max/min in the tensor dtype and converts to fp32: identical values
    if want_scale:
        coming = torch.maximum(mx, -mn)
        s = store["scales"]
        s[name] = torch.max(s[name], coming) if name in s else coming
    if want_shift:
        mid = (mx + mn) / 2
        s = store["shifts"]
        s[name] = 0.99 * s[name] + 0.01 * mid if name in s else mid


def _run(model, dataloader, num_samples, want_scale, want_shift):
    model.eval()
    device = next(model.parameters()).device
    store = {"scales": {}, "shifts": {}}
    hooks = []
    for name, m in model.named_modules():
        if isinstance(m, nn.Linear):
            hooks.append(m.register_forward_hook(functools.partial(_stat_input, store, want_scale, want_shift, name=name)))
    try:
        with torch.no_grad():
            for i in range(num_samples):
                model(dataloader[i][0].to(device))
    finally:
        for h in hooks:
            h.remove()
    to_host = lambda d: {k: v.float().cpu() for k, v in d.items()}
    return to_host(store["scales"]), to_host(store["shifts"])


def get_act_scales(model, dataloader, num_samples=128):
    """Running per-channel max of |x| over the inputs of every nn.Linear (generate_act_scale_shift.py:26-57)."""
    return _run(model, dataloader, num_samples, True, False)[0]


def get_act_shifts(model, dataloader, num_samples=128):
    """0.99/0.01 EMA of the per-channel (max+min)/2 of the inputs of every nn.Linear (:59-95)."""
    return _run(model, dataloader, num_samples, False, True)[1]


def get_act_scales_and_shifts(model, dataloader, num_samples=128):
    """Both dictionaries from a single pass over the calibration samples."""
    return _run(model, dataloader, num_samples, True, True)
