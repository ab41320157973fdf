"""Data-parallel plumbing for block calibration: one process per GPU, torch.distributed (backend
"nccl" = RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The path has exactly two exchange steps per optimizer step, both latency-bound small messages:
  * one all-reduce (SUM, then / world) of all LWC/LET(/n_bits) gradients packed in one flat fp32 buffer;
  * optionally, per activation-quantizer call, one MAX all-reduce of the packed [-min, max] channel
    statistics so that every rank derives the scale of the GLOBAL batch.  With the samples dealt out round-robin
    (sample_indices) the W ranks' step j then is batch j of the reference run with --batch_size W.
Sample shards and weight-row shards need no communication.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


@dataclass
class DPContext:
    group: Optional[object] = None
    rank: int = 0
    world: int = 1
    sync_stats: bool = False

    @classmethod
    def from_env(cls, sync_stats=False):
        if dist.is_available() and dist.is_initialized():
            return cls(dist.group.WORLD, dist.get_rank(), dist.get_world_size(), sync_stats)
        return cls(None, 0, 1, False)

    @property
    def active(self):
        return self.world > 1


def shard_range(n: int, rank: int, world: int):
    """Contiguous, balanced [start, stop) of n items for this rank (first n % world ranks get one more)."""
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def owner_of(index: int, n: int, world: int) -> int:
    base, extra = divmod(n, world)
    cut = extra * (base + 1)
    if index < cut:
        return index // (base + 1)
    return extra + (index - cut) // base if base else world - 1


def sample_indices(n: int, rank: int, world: int):
    """Calibration samples held by ``rank``: rank, rank + world, ...  Strided on purpose: the ranks' local step j then
    works on samples j * world .. j * world + world - 1, which is batch j of the reference run with
    --batch_size world (quantize/omniquant.py:502-510)."""
    return range(rank, n, world)


def sample_owner(index: int, world: int):
    """(rank, local position) of calibration sample ``index`` under sample_indices()."""
    return index % world, index // world


# Collectives.  On RCCL ("nccl") GPU tensors go to the collective as they are.  The gloo backend is what the CPU tests
# and the one-GPU rehearsals of N ranks use; its own path for GPU tensors stalled now and then when two ranks shared one
# GPU (both ranks inside the same all-reduce until the 60 s timeout: seen twice in three rehearsals of bench.py), so under
# gloo a GPU tensor makes the trip through a host copy.
def _staged(t: torch.Tensor, group) -> bool:
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce(t: torch.Tensor, op, group=None):
    if _staged(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


def broadcast(t: torch.Tensor, src: int, group=None):
    if _staged(t, group):
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)
    return t


def all_gather_into(out: torch.Tensor, part: torch.Tensor, group=None):
    if _staged(out, group):
        h = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h, part.cpu(), group=group)
        out.copy_(h)
    else:
        dist.all_gather_into_tensor(out, part, group=group)
    return out


class GradBucket:
    """One flat fp32 buffer with a view per parameter: packing and unpacking are one multi-tensor copy each."""

    def __init__(self, params: List[torch.nn.Parameter]):
        self.shapes = [(p.shape, p.dtype, p.device) for p in params]
        n = sum(p.numel() for p in params)
        self.flat = torch.empty(n, dtype=torch.float32, device=params[0].device)
        self.views, off = [], 0
        for p in params:
            self.views.append(self.flat[off:off + p.numel()].view(p.shape))
            off += p.numel()

    def fits(self, params):
        return len(params) == len(self.shapes) and all((p.shape, p.dtype, p.device) == s for p, s in zip(params, self.shapes))


def allreduce_mean_grads(params: Iterable[torch.nn.Parameter], ctx: DPContext, buf: Optional[GradBucket] = None):
    """Average the gradients of ``params`` across ranks with ONE collective on a packed fp32 buffer (pack, all-reduce,
    scale, unpack: four launches however many parameters there are).  Parameters without a gradient contribute
    zeros (and receive the mean of the others).  Returns the bucket for reuse by the next step."""
    if not ctx.active:
        return buf
    params = [p for p in params]
    if not params:
        return buf
    if buf is None or not buf.fits(params):
        buf = GradBucket(params)
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    grads = [p.grad for p in params]
    # The multi-tensor copy is only taken when no pair converts: on this PyTorch 2.10 / ROCm 7 build
    # torch._foreach_copy_ returns wrong values for the fp16 tensors of a list that mixes fp32 -> fp32 and
    # fp32 -> fp16 pairs, and for the tensors after them (tools/dev/foreach_copy_check.py).
    fused = all(g.dtype == torch.float32 for g in grads)
    if fused:
        torch._foreach_copy_(buf.views, grads)
    else:
        for v, g in zip(buf.views, grads):
            v.copy_(g)
    all_reduce(buf.flat, dist.ReduceOp.SUM, ctx.group)
    buf.flat.div_(ctx.world)
    if fused:
        torch._foreach_copy_(grads, buf.views)
    else:
        for g, v in zip(grads, buf.views):
            g.copy_(v)
    return buf


def allreduce_minmax(cmin: torch.Tensor, cmax: torch.Tensor, ctx: DPContext):
    """Global per-channel min/max with one MAX all-reduce of [-min, max] (float32 vectors, in place)."""
    if not ctx.active:
        return cmin, cmax
    packed = torch.cat((-cmin, cmax))
    all_reduce(packed, dist.ReduceOp.MAX, ctx.group)
    c = cmin.numel()
    cmin.copy_(-packed[:c])
    cmax.copy_(packed[c:])
    return cmin, cmax


def broadcast_from(t: torch.Tensor, src: int, ctx: DPContext):
    if ctx.active:
        broadcast(t, src, ctx.group)
    return t


def shard_weight_rows(weights: List[torch.Tensor], ctx: DPContext):
    """Output-channel shards of a list of [O, I] matrices for this rank: (row slice, view) pairs.
    Row-wise quantizers treat every output row independently, so shards need no exchange."""
    out = []
    for w in weights:
        a, b = shard_range(w.shape[0], ctx.rank, ctx.world)
        out.append((slice(a, b), w[a:b]))
    return out


def gather_row_shards(shard: torch.Tensor, rows_total: int, ctx: DPContext, out: Optional[torch.Tensor] = None):
    """Reassemble an [O, I] matrix on every rank from the ranks' row shards (the ranges of shard_range, in rank order).

    Equal shards are gathered straight into the result (its memory IS the concatenation of the shards: one collective,
    no copy); otherwise every rank pads its shard to the largest one, one collective fills a [world * max_rows, I]
    staging tensor and the valid rows are copied out.  Point-to-point xGMI links carry O * I * (world - 1) / world
    elements into every rank either way."""
    cols = shard.shape[1:]
    if out is None:
        out = torch.empty((rows_total, *cols), dtype=shard.dtype, device=shard.device)
    if not ctx.active:
        out.copy_(shard)
        return out
    a, b = shard_range(rows_total, ctx.rank, ctx.world)
    if shard.shape[0] != b - a:
        raise ValueError(f"rank {ctx.rank} holds {shard.shape[0]} rows, its range [{a}, {b}) has {b - a}")
    if rows_total % ctx.world == 0:
        all_gather_into(out, shard.contiguous(), ctx.group)
        return out
    most = -(-rows_total // ctx.world)
    padded = torch.zeros((most, *cols), dtype=shard.dtype, device=shard.device)
    padded[:b - a].copy_(shard)
    stage = torch.empty((ctx.world * most, *cols), dtype=shard.dtype, device=shard.device)
    all_gather_into(stage, padded, ctx.group)
    stage = stage.view(ctx.world, most, *cols)
    for r in range(ctx.world):
        ra, rb = shard_range(rows_total, r, ctx.world)
        out[ra:rb].copy_(stage[r, :rb - ra])
    return out


def quantize_rows_sharded(weights: List[torch.Tensor], ctx: DPContext, quantize, gather: bool = True):
    """The for-all-layers weight quantization sharded by output channel: ``quantize(row_shard)`` runs on this rank's
    rows of every [O, I] matrix (row-wise quantizers need nothing from other rows), and with ``gather`` every rank gets
    the full dequantized matrices back.  Returns the list of full matrices (gather) or of (row slice, shard) pairs."""
    out = []
    for sl, w in shard_weight_rows(weights, ctx):
        q = quantize(w)
        if gather:
            out.append(gather_row_shards(q, weights[len(out)].shape[0], ctx))
        else:
            out.append((sl, q))
    return out

