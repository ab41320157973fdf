"""RPTQ channel reorder: statistics hooks, reorder index and the R1..R5 rewiring of a decoder block.
Drop-in for the reference module (quantize/reorder_utils.py:1-149).

The running per-channel max/min live in two module-level dictionaries that are never cleared between
layers — the reference behaves the same way (layer i's index is computed from statistics accumulated
over layers 0..i because every block's modules carry the same names)."""
from __future__ import annotations

import numpy as np
import torch

from . import functional as F

oc_maxmin_dict = {}
ic_maxmin_dict = {}


def _channel_stats(t: torch.Tensor):
    """(max, min) per last-axis channel over all leading axes, in t's dtype, via the column-statistics kernel."""
    C = t.shape[-1]
    t2 = t.detach().contiguous().reshape(-1, C)
    st = F.act_calibrate(t2, 8, "minmax", F.ClusterLayout.get([], C, t.device))
    return st["chan_max"].to(t.dtype), st["chan_min"].to(t.dtype)


def _accumulate(store, name, xmax, xmin):
    if name not in store:
        store[name] = (xmax, xmin)
    else:
        store[name] = (torch.max(store[name][0], xmax), torch.min(store[name][1], xmin))


def layer_omax_hook(m, i, o):
    """Forward hook: running channel max/min of a module's OUTPUT (R1, R2, R4 statistics)."""
    if not isinstance(o, torch.Tensor) or o.ndim not in (2, 3):
        return
    xmax, xmin = _channel_stats(o)
    _accumulate(oc_maxmin_dict, m.name, xmax, xmin)


def layer_i0max_hook(m, i, o):
    """Forward hook: running channel max/min of a module's first INPUT (R3, R5 statistics)."""
    if len(i) == 0 or not isinstance(i[0], torch.Tensor) or i[0].ndim not in (2, 3):
        return
    xmax, xmin = _channel_stats(i[0])
    _accumulate(ic_maxmin_dict, m.name, xmax, xmin)


def peg_tensor_calc_reorder_index(xmax, xmin, n_clusters, n_heads=None):
    """Per head: channels sorted by ascending range, cut into n_clusters equal clusters.

    With list inputs only the first entry decides the order (the reference stacks [q, k] but reads
    column 0 only).  Returns (int64 index tensor [C], numpy int array of cluster sizes)."""
    if n_heads is None:
        n_heads = 1
    if isinstance(xmax, (list, tuple)):
        xmax, xmin = xmax[0], xmin[0]
    C = xmax.numel()
    per = C // n_heads
    index = F.reorder_index(xmax.reshape(-1), xmin.reshape(-1), n_heads)
    counts = np.hstack([np.array([per // n_clusters] * n_clusters) for _ in range(n_heads)])
    return index, counts


tensor_calc_reorder_index = peg_tensor_calc_reorder_index


def _select(t, dim, index):
    return torch.index_select(t, dim, index.to(t.device))


def R1_reorder(layer_norm, qproj, kproj, vproj, index, counts):
    layer_norm.register_buffer("reorder_index", index)
    layer_norm.weight.data = _select(layer_norm.weight.data, 0, index)
    layer_norm.bias.data = _select(layer_norm.bias.data, 0, index)
    for proj in (qproj, kproj, vproj):
        proj.act_quantizer.cluster_dim = 2
        proj.act_quantizer.cluster_counts = counts
    for proj in (qproj, kproj, vproj):
        proj.weight.data = _select(proj.weight.data, 1, index)
        proj.set_ic_cluster_counts(counts, a_dim=2)


def R2_reorder(qproj, kproj, qkt_matmul, index, counts):
    for proj in (qproj, kproj):
        proj.weight.data = _select(proj.weight.data, 0, index)
        proj.bias.data = _select(proj.bias.data, 0, index)
    qkt_matmul.set_ic_cluster_counts(counts, x1_dim=2, x2_dim=2)


def R3_reorder(vproj, pv_matmul, out_proj, index, counts):
    vproj.weight.data = _select(vproj.weight.data, 0, index)
    vproj.bias.data = _select(vproj.bias.data, 0, index)
    pv_matmul.set_ic_cluster_counts(counts, cluster_x1=False)
    out_proj.weight.data = _select(out_proj.weight.data, 1, index)
    out_proj.set_ic_cluster_counts(counts)


def R4_reorder(layer_norm, fc1, index, counts):
    layer_norm.register_buffer("reorder_index", index)
    layer_norm.weight.data = _select(layer_norm.weight.data, 0, index)
    layer_norm.bias.data = _select(layer_norm.bias.data, 0, index)
    fc1.act_quantizer.cluster_dim = 1  # the MLP runs on 2-D activations
    fc1.act_quantizer.cluster_counts = counts
    fc1.weight.data = _select(fc1.weight.data, 1, index)
    fc1.set_ic_cluster_counts(counts, a_dim=1)


def R5_reorder(fc1, fc2, index, counts):
    fc1.weight.data = _select(fc1.weight.data, 0, index)
    fc1.bias.data = _select(fc1.bias.data, 0, index)
    fc2.weight.data = _select(fc2.weight.data, 1, index)
    fc2.set_ic_cluster_counts(counts, a_dim=1)
