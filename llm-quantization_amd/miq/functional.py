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


def _flags(clamp_asym: bool) -> int:
    return 1 if clamp_asym else 0


class LwcSigmoid(torch.autograd.Function):
    """apply(f_0, f_1, ...) -> (sigmoid(f_0), sigmoid(f_1), ...) for learnable-clipping factors of one dtype on one
    device.  The kernels take ACTIVATED clip factors: the sigmoid is the routine the reference itself runs
    (``self.sigmoid(self.upbound_factor)``, quantize/quantizer.py:259-261 -- ATen's), so the clip factor, and with it
    the scale of every row, is the reference's bit for bit; the backward is ATen's sigmoid_backward on the gradients
    the kernels return for the activated factors.  All tensors go through ONE concatenated buffer: two launches each
    way however many quantizers a block holds."""

    @staticmethod
    def forward(ctx, *fs):
        sizes = [f.numel() for f in fs]
        flat = fs[0].reshape(-1) if len(fs) == 1 else torch.cat([f.reshape(-1) for f in fs])
        act = torch.sigmoid(flat)
        ctx.save_for_backward(act)
        ctx.sizes = sizes
        ctx.set_materialize_grads(False)
        return tuple(a.view(f.shape) for a, f in zip(act.split(sizes), fs))

    @staticmethod
    def backward(ctx, *gs):
        (act,) = ctx.saved_tensors
        if all(g is None for g in gs):
            return (None,) * len(gs)
        parts = [(g.reshape(-1) if g is not None else act.new_zeros(n)) for g, n in zip(gs, ctx.sizes)]
        gflat = parts[0] if len(parts) == 1 else torch.cat(parts)
        d = torch.ops.aten.sigmoid_backward(gflat.to(act.dtype), act)
        return tuple(v.view(g.shape) if g is not None else None for v, g in zip(d.split(ctx.sizes), gs))


def lwc_activate(factors):
    """[sigmoid(f) for f in factors] through LwcSigmoid (differentiable; factors of one dtype and device)."""
    factors = list(factors)
    if not factors:
        return []
    return list(LwcSigmoid.apply(*factors))


def rowquant_uniform_fwd(x, n_bits: int, symmetric: bool, up=None, low=None, group: int = 0,
                         row_bits: Optional[torch.Tensor] = None, clamp_asym: bool = False,
                         want_codes: bool = False):
    """Returns (out, scale[rows,1], rzp[rows,1], codes|None).

    ``up`` / ``low`` hold the ACTIVATED clip factors sigmoid(factor) (lwc_activate); gradients come back for those.
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
    with L.on_device(dev):
        L.call("miq_rowquant_uniform_fwd", L.ptr(x), L.ptr(up), L.ptr(low), L.ptr(row_bits), L.ptr(out),
               L.ptr(scale), L.ptr(rzp), L.ptr(codes), L.i64(O), L.i64(I), L.i64(g), L.i32(int(n_bits)),
               L.i32(int(bool(symmetric))), L.i32(_flags(clamp_asym)), L.i32(L.dtype_code(x)),
               L.stream_of(x))
    return out, scale, rzp, codes


def rowquant_uniform_bwd(grad_out, x, n_bits: int, symmetric: bool, up=None, low=None, group: int = 0,
                         row_bits=None, clamp_asym: bool = False, want_dx: bool = True):
    """Returns (dx|None, d_up|None, d_low|None)."""
    dev = L.check_gpu(grad_out, x, up, low, row_bits)
    O, I, g, rows = _rows_of(x, group, up is not None)
    dx = torch.empty_like(x) if want_dx else None
    d_up = torch.empty_like(up) if up is not None else None
    d_low = torch.empty_like(low) if low is not None else None
    with L.on_device(dev):
        L.call("miq_rowquant_uniform_bwd", L.ptr(grad_out), L.ptr(x), L.ptr(up), L.ptr(low), L.ptr(row_bits),
               L.ptr(dx), L.ptr(d_up), L.ptr(d_low), L.i64(O), L.i64(I), L.i64(g), L.i32(int(n_bits)),
               L.i32(int(bool(symmetric))), L.i32(_flags(clamp_asym)), L.i32(L.dtype_code(x)),
               L.stream_of(x))
    return dx, d_up, d_low


def _ptr_table(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[0 if t is None else t.data_ptr() for t in tensors])


def _check_batch(xs, ups, lows, gs=None):
    """The batched row kernels read raw pointer tables: everything they assume about the tensors behind them."""
    if not xs or len(xs) > 16:
        raise L.MiqError(f"a batch holds 1..16 matrices, got {len(xs)}")
    if (ups is None) != (lows is None):
        raise L.MiqError("LWC factors come in pairs (up and low)")
    I, dt = xs[0].shape[-1], xs[0].dtype
    for x in xs:
        if x.dim() != 2 or x.shape[1] != I or x.dtype != dt or not x.is_contiguous():
            raise L.MiqError("batched row quantization needs contiguous 2-D matrices of one width and dtype")
    for name, fs in (("up", ups), ("low", lows)):
        if fs is None:
            continue
        if len(fs) != len(xs):
            raise L.MiqError(f"{len(fs)} {name} factors for {len(xs)} matrices")
        for f, x in zip(fs, xs):
            if f.numel() != x.shape[0] or f.dtype != dt or not f.is_contiguous():
                raise L.MiqError(f"every {name} factor must be a contiguous tensor of {x.shape[0]} elements of dtype {dt}")
    if gs is not None:
        if len(gs) != len(xs):
            raise L.MiqError(f"{len(gs)} gradients for {len(xs)} matrices")
        for g, x in zip(gs, xs):
            if g.shape != x.shape or g.dtype != dt or not g.is_contiguous():
                raise L.MiqError("every gradient must be a contiguous tensor of its matrix's shape and dtype")


def rowquant_uniform_fwd_batch(xs, n_bits: int, symmetric: bool, ups=None, lows=None):
    """Row-wise uniform fake quantization of several [O_i, I] matrices of one width and dtype in one launch.
    Returns (outs, scales, rzps) lists."""
    import ctypes
    lwc = ups is not None
    dev = L.check_gpu(*xs, *(ups or []), *(lows or []))
    _check_batch(xs, ups, lows)
    I = xs[0].shape[1]
    outs = [torch.empty_like(x) for x in xs]
    scales = [torch.empty((x.shape[0], 1), dtype=x.dtype, device=dev) for x in xs]
    rzps = [torch.empty_like(s) for s in scales]
    O = (ctypes.c_int64 * len(xs))(*[x.shape[0] for x in xs])
    with L.on_device(dev):
        L.call("miq_rowquant_uniform_fwd_batch", L.i32(len(xs)), _ptr_table(xs), _ptr_table(ups) if lwc else None,
               _ptr_table(lows) if lwc else None, _ptr_table(outs), _ptr_table(scales), _ptr_table(rzps), O, L.i64(I),
               L.i32(int(n_bits)), L.i32(int(bool(symmetric))), L.i32(_flags(False)), L.i32(L.dtype_code(xs[0])),
               L.stream_of(xs[0]))
    return outs, scales, rzps


def rowquant_uniform_bwd_batch(gs, xs, n_bits: int, symmetric: bool, ups=None, lows=None, want_dx: bool = True):
    """Backward of the above.  Returns (dxs|None, d_ups|None, d_lows|None) lists."""
    import ctypes
    lwc = ups is not None
    dev = L.check_gpu(*gs, *xs, *(ups or []), *(lows or []))
    _check_batch(xs, ups, lows, gs)
    I = xs[0].shape[1]
    dxs = [torch.empty_like(x) for x in xs] if want_dx else None
    d_ups = [torch.empty_like(u) for u in ups] if lwc else None
    d_lows = [torch.empty_like(u) for u in lows] if lwc else None
    O = (ctypes.c_int64 * len(xs))(*[x.shape[0] for x in xs])
    none = [None] * len(xs)
    with L.on_device(dev):
        L.call("miq_rowquant_uniform_bwd_batch", L.i32(len(xs)), _ptr_table(gs), _ptr_table(xs), _ptr_table(ups) if lwc else None,
               _ptr_table(lows) if lwc else None, _ptr_table(dxs) if want_dx else None, _ptr_table(d_ups if lwc else none),
               _ptr_table(d_lows if lwc else none), O, L.i64(I), L.i32(int(n_bits)), L.i32(int(bool(symmetric))),
               L.i32(_flags(False)), L.i32(L.dtype_code(xs[0])), L.stream_of(xs[0]))
    return dxs, d_ups, d_lows


def _check_row_bits(xs, row_bits):
    if len(row_bits) != len(xs):
        raise L.MiqError(f"{len(row_bits)} bit-width vectors for {len(xs)} matrices")
    for x, rb in zip(xs, row_bits):
        if rb.dtype != torch.float32 or rb.numel() != x.shape[0] or not rb.is_contiguous():
            raise L.MiqError(f"every bit-width vector must be a contiguous float32 tensor of {x.shape[0]} entries")


def rowquant_mix2_fwd_batch(xs, row_bits, symmetric: bool, ups=None, lows=None, clamp_asym: bool = True):
    """Row-wise uniform fake quantization with per-row integer bit-widths (the fixed mixed-bit quantizer) of several
    [O_i, I] matrices of one width and dtype in one launch.  Returns (outs, scales, rzps) lists; bit-identical to
    rowquant_uniform_fwd(..., row_bits=...) per matrix."""
    import ctypes
    lwc = ups is not None
    dev = L.check_gpu(*xs, *row_bits, *(ups or []), *(lows or []))
    _check_batch(xs, ups, lows)
    _check_row_bits(xs, row_bits)
    outs = [torch.empty_like(x) for x in xs]
    scales = [torch.empty((x.shape[0], 1), dtype=x.dtype, device=dev) for x in xs]
    rzps = [torch.empty_like(s) for s in scales]
    O = (ctypes.c_int64 * len(xs))(*[x.shape[0] for x in xs])
    with L.on_device(dev):
        L.call("miq_rowquant_mix2_fwd_batch", L.i32(len(xs)), _ptr_table(xs), _ptr_table(ups) if lwc else None,
               _ptr_table(lows) if lwc else None, _ptr_table(row_bits), _ptr_table(outs), _ptr_table(scales), _ptr_table(rzps), O,
               L.i64(xs[0].shape[1]), L.i32(int(bool(symmetric))), L.i32(_flags(clamp_asym)),
               L.i32(L.dtype_code(xs[0])), L.stream_of(xs[0]))
    return outs, scales, rzps


def rowquant_mix2_bwd_batch(gs, xs, row_bits, symmetric: bool, ups=None, lows=None, clamp_asym: bool = True,
                            want_dx: bool = True):
    """Backward of the above.  Returns (dxs|None, d_ups|None, d_lows|None) lists."""
    import ctypes
    lwc = ups is not None
    dev = L.check_gpu(*gs, *xs, *row_bits, *(ups or []), *(lows or []))
    _check_batch(xs, ups, lows, gs)
    _check_row_bits(xs, row_bits)
    dxs = [torch.empty_like(x) for x in xs] if want_dx else None
    d_ups = [torch.empty_like(u) for u in ups] if lwc else None
    d_lows = [torch.empty_like(u) for u in lows] if lwc else None
    O = (ctypes.c_int64 * len(xs))(*[x.shape[0] for x in xs])
    none = [None] * len(xs)
    with L.on_device(dev):
        L.call("miq_rowquant_mix2_bwd_batch", L.i32(len(xs)), _ptr_table(gs), _ptr_table(xs), _ptr_table(ups) if lwc else None,
               _ptr_table(lows) if lwc else None, _ptr_table(row_bits), _ptr_table(dxs) if want_dx else None,
               _ptr_table(d_ups if lwc else none), _ptr_table(d_lows if lwc else none), O, L.i64(xs[0].shape[1]),
               L.i32(int(bool(symmetric))), L.i32(_flags(clamp_asym)), L.i32(L.dtype_code(xs[0])),
               L.stream_of(xs[0]))
    return dxs, d_ups, d_lows


class UniformRowFQBatch(torch.autograd.Function):
    """apply(n_bits, symmetric, lwc, x_0, [up_0, low_0,] x_1, ...) -> (out_0, scale_0, rzp_0, out_1, ...):
    several weight matrices of one width through one forward and one backward launch."""

    @staticmethod
    def forward(ctx, n_bits, symmetric, lwc, *ts):
        step = 3 if lwc else 1
        xs = [t.contiguous() for t in ts[0::step]]
        ups = list(ts[1::step]) if lwc else None
        lows = list(ts[2::step]) if lwc else None
        outs, scales, rzps = rowquant_uniform_fwd_batch(xs, n_bits, symmetric, ups, lows)
        ctx.cfg = (n_bits, symmetric, lwc, len(xs))
        ctx.save_for_backward(*xs, *(ups or []), *(lows or []))
        ret = []
        for o, s, z in zip(outs, scales, rzps):
            ret += [o, s, z]
        ctx.mark_non_differentiable(*scales, *rzps)
        ctx.set_materialize_grads(False)  # no zero-filled gradients for the scale / zero-point outputs
        return tuple(ret)

    @staticmethod
    def backward(ctx, *grads):
        n_bits, symmetric, lwc, n = ctx.cfg
        saved = ctx.saved_tensors
        xs = list(saved[:n])
        ups = list(saved[n:2 * n]) if lwc else None
        lows = list(saved[2 * n:3 * n]) if lwc else None
        gs = [(g if g is not None else torch.zeros_like(x)).contiguous() for g, x in zip(grads[0::3], xs)]
        step = 3 if lwc else 1
        need_dx = any(ctx.needs_input_grad[3 + i * step] for i in range(n))
        dxs, d_ups, d_lows = rowquant_uniform_bwd_batch(gs, xs, n_bits, symmetric, ups, lows, want_dx=need_dx)
        out = [None, None, None]
        for i in range(n):
            out.append(dxs[i] if need_dx else None)
            if lwc:
                out += [d_ups[i], d_lows[i]]
        return tuple(out)


class Mix2RowFQBatch(torch.autograd.Function):
    """apply(symmetric, lwc, row_bits, x_0, [up_0, low_0,] x_1, ...) -> (out_0, out_1, ...): several weight matrices of one
    width through one forward and one backward launch of the fixed mixed-bit quantizer; row_bits is the list of the
    matrices' float32 per-row bit-width vectors (not differentiable)."""

    @staticmethod
    def forward(ctx, symmetric, lwc, row_bits, *ts):
        step = 3 if lwc else 1
        xs = [t.contiguous() for t in ts[0::step]]
        ups = list(ts[1::step]) if lwc else None
        lows = list(ts[2::step]) if lwc else None
        outs, _scales, _rzps = rowquant_mix2_fwd_batch(xs, row_bits, symmetric, ups, lows)
        ctx.cfg = (symmetric, lwc, len(xs), list(row_bits))
        ctx.save_for_backward(*xs, *(ups or []), *(lows or []))
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        symmetric, lwc, n, row_bits = ctx.cfg
        saved = ctx.saved_tensors
        xs = list(saved[:n])
        ups = list(saved[n:2 * n]) if lwc else None
        lows = list(saved[2 * n:3 * n]) if lwc else None
        gs = [(g if g is not None else torch.zeros_like(x)).contiguous() for g, x in zip(grads, xs)]
        step = 3 if lwc else 1
        need_dx = any(ctx.needs_input_grad[3 + i * step] for i in range(n))
        dxs, d_ups, d_lows = rowquant_mix2_bwd_batch(gs, xs, row_bits, symmetric, ups, lows, want_dx=need_dx)
        out = [None, None, None]
        for i in range(n):
            out.append(dxs[i] if need_dx else None)
            if lwc:
                out += [d_ups[i], d_lows[i]]
        return tuple(out)


class UniformRowFQ(torch.autograd.Function):
    """out, scale, rzp = fq(x, up, low); gradients for x, up, low."""

    @staticmethod
    def forward(ctx, x, up, low, n_bits, symmetric, group, row_bits, clamp_asym):
        x = x.contiguous()
        out, scale, rzp, _ = rowquant_uniform_fwd(x, n_bits, symmetric, up, low, group, row_bits, clamp_asym)
        ctx.save_for_backward(x, up, low, row_bits)
        ctx.cfg = (n_bits, symmetric, group, clamp_asym)
        ctx.mark_non_differentiable(scale, rzp)
        ctx.set_materialize_grads(False)  # no zero-filled gradients for the scale / zero-point outputs
        return out, scale, rzp

    @staticmethod
    def backward(ctx, g, _gs, _gz):
        x, up, low, row_bits = ctx.saved_tensors
        n_bits, symmetric, group, clamp_asym = ctx.cfg
        need_dx = ctx.needs_input_grad[0]
        need_p = up is not None and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        if g is None or not (need_dx or need_p):
            return (None,) * 8
        dx, d_up, d_low = rowquant_uniform_bwd(g.contiguous(), x, n_bits, symmetric, up, low, group, row_bits,
                                               clamp_asym, want_dx=need_dx)
        return dx, d_up, d_low, None, None, None, None, None


# ---------------------------------------------------------------------------------------------
# static activation quantisation over the last axis
# ---------------------------------------------------------------------------------------------
class ClusterLayout:
    """Contiguous channel clusters of a C-channel axis: device prefix sums + bookkeeping.

    ``counts`` None means one cluster spanning all channels (per-tensor statistics).
    Channels beyond sum(counts) belong to no cluster and keep per-channel statistics
    (quantize/quantizer.py:201-210 only visits the listed clusters).
    """

    _cache = {}

    def __init__(self, counts, C: int, device):
        # per-tensor statistics are one reduction over all elements: tied extremes share the scale gradient evenly,
        # whereas the extremes of an explicit cluster are found per channel first (ties split per channel, then per row)
        self.flat = counts is None
        if counts is None:
            counts = [C]
        counts = [int(c) for c in counts]
        if any(c < 0 for c in counts) or sum(counts) > C:
            raise L.MiqError(f"cluster sizes {counts[:4]}... do not fit {C} channels")
        if torch.device(device).type == "cuda" and torch.cuda.is_current_stream_capturing():
            # the prefix sums travel to the device by a copy from pageable host memory, which HIP refuses on a capturing
            # stream (hipErrorStreamCaptureUnsupported) and which invalidates the whole capture
            raise L.MiqError("a cluster layout is built from host lists and cannot be created during graph capture: run the "
                             "captured calls once eagerly first (ClusterLayout.get caches the layout)")
        starts = [0]
        for c in counts:
            starts.append(starts[-1] + c)
        self.n_clusters = len(counts)
        self.covered = starts[-1]
        self.C = C
        self.starts = torch.tensor(starts, dtype=torch.int32, device=device)
        # equal-sized clusters (every layout the reorder stages produce) are addressed by arithmetic; anything else
        # through a channel -> statistics-slot map (cluster id, or n_clusters + (c - covered) outside every cluster)
        self.cluster_size = counts[0] if counts and counts[0] > 0 and all(c == counts[0] for c in counts) else 0
        self._counts = counts
        self._device = device
        self._slot_of = None

    @property
    def slot_of(self):
        """The channel -> slot map of layouts whose slots are not found by arithmetic (None for equal clusters)."""
        return None if self.cluster_size else self.slot_map()

    def slot_map(self):
        if self._slot_of is None:
            slot = []
            for k, n in enumerate(self._counts):
                slot += [k] * n
            slot += [self.n_clusters + i for i in range(self.C - self.covered)]
            self._slot_of = torch.tensor(slot, dtype=torch.int32, device=self._device)
        return self._slot_of

    @classmethod
    def get(cls, counts, C: int, device):
        key = (None if counts is None else tuple(int(c) for c in counts), int(C), str(device))
        hit = cls._cache.get(key)
        if hit is None:
            if len(cls._cache) > 256:
                cls._cache.clear()
            hit = cls._cache[key] = cls(counts, C, device)
        return hit


def colstat_chunks(R: int, C: int, dtype) -> int:
    return int(L.lib().miq_colstat_chunks(L.i64(R), L.i64(C), L.i32(L._DT[dtype])))


def act_calibrate(x2d, n_bits: int, metric: str, layout: ClusterLayout, cached_min=None, cached_max=None, idx=None,
                  dp=None):
    """Statistics pass + cluster merge + scale.  x2d is [R, C] contiguous.

    Returns dict(scale, zp, rzp, cmin, cmax: [C] in x's dtype; chan_min, chan_max: [C] float32
    per-channel statistics before the cluster merge, needed by the backward pass).
    ``dp``: a miq.dist.DPContext with sync_stats=True makes the statistics those of the global batch
    (one MAX all-reduce of the packed [-min, max] vectors).
    """
    dev = L.check_gpu(x2d, cached_min, cached_max, idx)
    R, C = x2d.shape
    nchunk = colstat_chunks(R, C, x2d.dtype)
    part = torch.empty((2, nchunk + 1, C), dtype=torch.float32, device=dev)
    outs = torch.empty((5, C), dtype=x2d.dtype, device=dev)
    if idx is not None and idx.dtype != torch.int32:
        raise L.MiqError("reorder index must be int32 on the device")
    code = L.dtype_code(x2d)
    st = L.stream_of(x2d)
    with L.on_device(dev):
        L.call("miq_colstat_minmax", L.ptr(x2d), L.i64(R), L.i64(C), L.ptr(part[0]), L.ptr(part[1]), L.i32(nchunk),
               L.i32(code), st)
        fold = nchunk
        if dp is not None and dp.active and dp.sync_stats:
            from .dist import allreduce_minmax
            cm = part[0, :nchunk].amin(0)
            cx = part[1, :nchunk].amax(0)
            if idx is not None:
                cm, cx = cm[idx.long()], cx[idx.long()]
                idx = None
            allreduce_minmax(cm, cx, dp)
            part[0, 0].copy_(cm)
            part[1, 0].copy_(cx)
            fold = 1
        L.call("miq_cluster_scale", L.ptr(part[0]), L.ptr(part[1]), L.i32(fold), L.i64(C), L.ptr(idx),
               L.ptr(cached_min), L.ptr(cached_max), L.i32(1 if metric == "ema_minmax" else 0), L.ptr(layout.starts),
               L.i32(layout.n_clusters), L.i64(layout.covered), L.i32(int(n_bits)), L.ptr(outs[0]), L.ptr(outs[1]),
               L.ptr(outs[2]), L.ptr(outs[3]), L.ptr(outs[4]), L.i32(code), st)
    return {"cmin": outs[0], "cmax": outs[1], "scale": outs[2], "zp": outs[3], "rzp": outs[4],
            "chan_min": part[0, fold], "chan_max": part[1, fold]}


def channel_minmax(x2d):
    """Per-channel (min, max) over the rows of an [R, C] slab, fp32 [C] each (one streaming pass)."""
    dev = L.check_gpu(x2d)
    R, C = x2d.shape
    nchunk = colstat_chunks(R, C, x2d.dtype)
    part = torch.empty((2, nchunk, C), dtype=torch.float32, device=dev)
    with L.on_device(dev):
        L.call("miq_colstat_minmax", L.ptr(x2d), L.i64(R), L.i64(C), L.ptr(part[0]), L.ptr(part[1]), L.i32(nchunk),
               L.i32(L.dtype_code(x2d)), L.stream_of(x2d))
    return part[0].amin(0), part[1].amax(0)


def row_absmean_accum(g2d, acc):
    """acc[o] += mean_i |g[o, i]| (fp32 accumulator of O entries, updated in place)."""
    dev = L.check_gpu(g2d, acc)
    O, I = g2d.shape
    if acc.dtype != torch.float32 or acc.numel() != O or not acc.is_contiguous():
        raise L.MiqError("accumulator must be a contiguous float32 vector with one entry per row")
    with L.on_device(dev):
        L.call("miq_row_absmean_accum", L.ptr(g2d), L.ptr(acc), L.i64(O), L.i64(I), L.i32(L.dtype_code(g2d)), L.stream_of(g2d))
    return acc


def colquant_fwd(x2d, scale, rzp, n_bits: int, want_codes: bool = False):
    dev = L.check_gpu(x2d, scale, rzp)
    R, C = x2d.shape
    out = torch.empty_like(x2d)
    codes = torch.empty((R, C), dtype=torch.uint8, device=dev) if want_codes else None
    with L.on_device(dev):
        L.call("miq_colquant_fwd", L.ptr(x2d), L.ptr(scale), L.ptr(rzp), L.ptr(out), L.ptr(codes), L.i64(R), L.i64(C),
               L.i32(int(n_bits)), L.i32(L.dtype_code(x2d)), L.stream_of(x2d))
    return out, codes


def act_calib_quant_fwd(x2d, n_bits: int, metric: str, layout: ClusterLayout, cached_min=None, cached_max=None,
                        want_codes: bool = False):
    """Statistics + cluster merge + scale + quantize-dequantize of an [R, C] slab through one library
    call (miq_act_calib_quant_fwd).  Returns (out, codes|None, stats) with act_calibrate's stats dict."""
    dev = L.check_gpu(x2d, cached_min, cached_max)
    R, C = x2d.shape
    code = L.dtype_code(x2d)
    rows = colstat_chunks(R, C, x2d.dtype) + 1
    ws = torch.empty((2, rows, C), dtype=torch.float32, device=dev)
    outs = torch.empty((5, C), dtype=x2d.dtype, device=dev)
    out = torch.empty_like(x2d)
    codes = torch.empty((R, C), dtype=torch.uint8, device=dev) if want_codes else None
    with L.on_device(dev):
        L.call("miq_act_calib_quant_fwd", L.ptr(x2d), L.ptr(cached_min), L.ptr(cached_max),
               L.i32(1 if metric == "ema_minmax" else 0), L.ptr(layout.starts), L.i32(layout.n_clusters),
               L.i64(layout.covered), L.i32(int(n_bits)), L.ptr(ws), L.ptr(out), L.ptr(codes), L.ptr(outs[0]), L.ptr(outs[1]),
               L.ptr(outs[2]), L.ptr(outs[3]), L.ptr(outs[4]), L.i64(R), L.i64(C), L.i32(code), L.stream_of(x2d))
    stats = {"cmin": outs[0], "cmax": outs[1], "scale": outs[2], "zp": outs[3], "rzp": outs[4],
             "chan_min": ws[0, rows - 1], "chan_max": ws[1, rows - 1]}
    return out, codes, stats


_FRESH_WS = {}


def _fresh_workspace(dev, nbytes: int):
    """The statistics table of miq_act_fresh_quant_fwd: zero-filled once, then only ever handed to that function.
    One per (device, stream): calls that share a table must be ordered on one stream.  The table is never created or
    grown while its stream is being captured into a HIP graph: its zero fill would become a graph node (re-run by every
    replay, never run for later eager calls on that stream) and its memory would belong to the graph's private pool."""
    key = (dev.index, torch._C._cuda_getCurrentRawStream(dev.index))
    ws = _FRESH_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise L.MiqError("the activation quantizer's statistics table for this stream does not exist yet (or is too small) "
                             "and cannot be allocated during graph capture: run the same calls once eagerly ON THE CAPTURE "
                             "STREAM first (torch.cuda.graph(g, stream=s) after a warm-up under torch.cuda.stream(s))")
        ws = _FRESH_WS[key] = torch.zeros(max(nbytes, 1 << 16), dtype=torch.uint8, device=dev)
    return ws


def stats_of(vec):
    """The [5, C] block of the fresh-statistics forward as the dictionary the quantizer modules publish."""
    return {"cmin": vec[0], "cmax": vec[1], "scale": vec[2], "zp": vec[3], "rzp": vec[4], "vec": vec}


def act_fresh_quant_fwd(xs, n_bits: int, layout: ClusterLayout, want_codes: bool = False):
    """Calibrate (no cached statistics) + quantize up to 8 [R, C] slabs of one shape, dtype and cluster layout in
    two launches (miq_act_fresh_quant_fwd).  Returns (outs, codes | None, vecs): vecs[i] is the [5, C] block
    (merged min, merged max, scale, zero point, rounded zero point) of slab i."""
    dev = L.check_gpu(*xs)
    R, C = xs[0].shape
    for x in xs:
        if x.shape != xs[0].shape or x.dtype != xs[0].dtype:
            raise L.MiqError("slabs of one call must share shape and dtype")
    n = len(xs)
    outs = [torch.empty_like(x) for x in xs]
    codes = [torch.empty((R, C), dtype=torch.uint8, device=dev) for _ in xs] if want_codes else None
    vecs = torch.empty((n, 5, C), dtype=xs[0].dtype, device=dev)
    if R == 0 or C == 0:
        return outs, codes, list(vecs)
    nbytes = int(L.lib().miq_act_fresh_workspace(L.i32(n), L.i64(C), L.i32(layout.n_clusters), L.i64(layout.covered)))
    with L.on_device(dev):
        ws = _fresh_workspace(dev, nbytes)
        L.call("miq_act_fresh_quant_fwd", L.i32(n), _ptr_table(xs), L.i64(layout.cluster_size), L.ptr(layout.slot_of),
               L.i32(layout.n_clusters), L.i64(layout.covered), L.i32(int(n_bits)), L.ptr(ws), _ptr_table(outs),
               _ptr_table(codes) if want_codes else None, _ptr_table(list(vecs)), L.i64(R), L.i64(C),
               L.i32(L.dtype_code(xs[0])), L.stream_of(xs[0]))
    return outs, codes, list(vecs)


def act_fresh_quant_bwd(gs, xs, vecs, n_bits: int, layout: ClusterLayout):
    """Backward of act_fresh_quant_fwd for the same slabs (miq_act_fresh_quant_bwd).  Returns the list of dx."""
    dev = L.check_gpu(*gs, *xs, *vecs)
    R, C = xs[0].shape
    n = len(xs)
    dxs = [torch.empty_like(x) for x in xs]
    if R == 0 or C == 0:
        return dxs
    code = L.dtype_code(xs[0])
    nbytes = int(L.lib().miq_act_fresh_bwd_workspace(L.i32(n), L.i64(R), L.i64(C), L.i32(code)))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with L.on_device(dev):
        L.call("miq_act_fresh_quant_bwd", L.i32(n), _ptr_table(gs), _ptr_table(xs), _ptr_table(vecs), L.i64(layout.cluster_size),
               L.ptr(layout.starts), L.i32(layout.n_clusters), L.i64(layout.covered), L.i32(int(n_bits)), L.ptr(ws),
               _ptr_table(dxs), L.i64(R), L.i64(C), L.i32(int(layout.flat)), L.i32(code), L.stream_of(xs[0]))
    return dxs


def colquant_bwd(g2d, x2d, scale, rzp, n_bits: int, route=None, dp=None):
    """route = (chan_min, chan_max, kmin, kmax, layout) enables the min/max routing (calibration mode).
    With a stats-synchronised DPContext the per-channel sums are all-reduced between the two phases."""
    dev = L.check_gpu(g2d, x2d, scale, rzp)
    R, C = x2d.shape
    dx = torch.empty_like(x2d)
    nchunk = colstat_chunks(R, C, x2d.dtype)
    flat = L.i32(0)
    if route is not None:
        chan_min, chan_max, kmin, kmax, layout = route
        flat = L.i32(int(layout.flat))
        ws = torch.empty((3 * nchunk + 5) * C, dtype=torch.float32, device=dev)
        args = (L.ptr(chan_min), L.ptr(chan_max), L.ptr(kmin), L.ptr(kmax), L.ptr(layout.starts),
                L.i32(layout.n_clusters), L.i64(layout.covered), L.ptr(ws))
    else:
        args = (L.ptr(None),) * 5 + (L.i32(0), L.i64(0), L.ptr(None))
    tail = (L.i32(nchunk), L.ptr(dx), L.i64(R), L.i64(C), L.i32(int(n_bits)))
    code, st = L.i32(L.dtype_code(x2d)), L.stream_of(x2d)
    with L.on_device(dev):
        if route is not None and dp is not None and dp.active and dp.sync_stats:
            import torch.distributed as tdist
            from .dist import all_reduce
            L.call("miq_colquant_bwd", L.ptr(g2d), L.ptr(x2d), L.ptr(scale), L.ptr(rzp), *args, *tail, L.i32(1), flat, code, st)
            all_reduce(ws[3 * nchunk * C:(3 * nchunk + 3) * C], tdist.ReduceOp.SUM, dp.group)  # [dS, #max, #min] per channel
            L.call("miq_colquant_bwd", L.ptr(g2d), L.ptr(x2d), L.ptr(scale), L.ptr(rzp), *args, *tail, L.i32(2), flat, code, st)
        else:
            L.call("miq_colquant_bwd", L.ptr(g2d), L.ptr(x2d), L.ptr(scale), L.ptr(rzp), *args, *tail, L.i32(0), flat, code, st)
    return dx


def gather_lastdim(x, idx_i32):
    """out[..., c] = x[..., idx[c]] (torch.index_select along the last axis), idx int32 on the device."""
    dev = L.check_gpu(x, idx_i32)
    C = x.shape[-1]
    if idx_i32.dtype != torch.int32 or idx_i32.numel() != C:
        raise L.MiqError("gather index must be an int32 tensor with one entry per channel")
    R = x.numel() // C if C else 0
    out = torch.empty_like(x)
    with L.on_device(dev):
        L.call("miq_gather_lastdim", L.ptr(x), L.ptr(idx_i32), L.ptr(out), L.i64(R), L.i64(C), L.i32(L.dtype_code(x)),
               L.stream_of(x))
    return out


def gather_lastdim_batch(xs, idxs):
    """[gather_lastdim(x, idx) for x, idx in zip(xs, idxs)] for up to 8 tensors of one shape and dtype in one launch."""
    import ctypes
    dev = L.check_gpu(*xs, *idxs)
    C = xs[0].shape[-1]
    for x, i in zip(xs, idxs):
        if x.shape != xs[0].shape or x.dtype != xs[0].dtype:
            raise L.MiqError("tensors of one batched gather must share shape and dtype")
        if i.dtype != torch.int32 or i.numel() != C:
            raise L.MiqError("gather index must be an int32 tensor with one entry per channel")
    if len(idxs) != len(xs) or not 1 <= len(xs) <= 8:
        raise L.MiqError(f"a batched gather takes 1..8 tensors with one index each, got {len(xs)} and {len(idxs)}")
    R = xs[0].numel() // C if C else 0
    outs = [torch.empty_like(x) for x in xs]
    idx_tab = (ctypes.c_void_p * len(idxs))(*[i.data_ptr() for i in idxs])
    with L.on_device(dev):
        L.call("miq_gather_lastdim_batch", L.i32(len(xs)), _ptr_table(xs), idx_tab, _ptr_table(outs), L.i64(R), L.i64(C),
               L.i32(L.dtype_code(xs[0])), L.stream_of(xs[0]))
    return outs


class GatherLastDim(torch.autograd.Function):
    """index_select along the last axis by a permutation; the backward is the gather by the inverse."""

    @staticmethod
    def forward(ctx, x, idx_i32, inv_i32):
        ctx.save_for_backward(inv_i32)
        return gather_lastdim(x.contiguous(), idx_i32)

    @staticmethod
    def backward(ctx, g):
        (inv,) = ctx.saved_tensors
        return gather_lastdim(g.contiguous(), inv), None, None


def gather_layernorm_supported(x, weight, bias) -> bool:
    C = x.shape[-1]
    # hidden sizes only: for wider rows (measured at 11008 fp16) gather + ATen's LayerNorm is as fast
    ok = x.is_cuda and x.dtype in (torch.float16, torch.float32) and C % 4 == 0 and C <= 8192
    for t in (weight, bias):
        ok = ok and (t is None or (t.is_cuda and t.dtype in (torch.float16, torch.float32) and t.numel() == C))
    if weight is not None and bias is not None:
        ok = ok and weight.dtype == bias.dtype
    return bool(ok)


def gather_layernorm_fwd(x, idx_i32, weight, bias, eps: float, out_dtype=None):
    """(y, mean, rstd): y = layer_norm(index_select(x, -1, idx)) in one pass; mean / rstd fp32 [..., 1]."""
    dev = L.check_gpu(x, idx_i32, weight, bias)
    C = x.shape[-1]
    R = x.numel() // C if C else 0
    out = torch.empty_like(x, dtype=out_dtype or x.dtype)
    stat_shape = tuple(x.shape[:-1]) + (1,)
    mean = torch.empty(stat_shape, dtype=torch.float32, device=dev)
    rstd = torch.empty(stat_shape, dtype=torch.float32, device=dev)
    wb = weight if weight is not None else bias
    w_code = L._DT[wb.dtype] if wb is not None else L.F32
    with L.on_device(dev):
        L.call("miq_gather_layernorm_fwd", L.ptr(x), L.ptr(idx_i32), L.ptr(weight), L.ptr(bias), L.ptr(out), L.ptr(mean),
               L.ptr(rstd), L.i64(R), L.i64(C), L.f32(float(eps)), L.i32(L.dtype_code(x)), L.i32(w_code),
               L.i32(L.dtype_code(out)), L.stream_of(x))
    return out, mean, rstd


def gather_layernorm_bwd(gy, x, idx_i32, weight, mean, rstd, need_dx=True, need_dw=False, need_db=False):
    """(dx | None, d_weight | None, d_bias | None) of layer_norm(index_select(x, -1, idx), weight, bias) from the saved
    row statistics (miq_gather_layernorm_bwd); d_weight / d_bias are fp32 [C]."""
    gy = gy.contiguous()
    dev = L.check_gpu(gy, x, idx_i32, weight, mean, rstd)
    C = x.shape[-1]
    R = x.numel() // C
    dx = torch.empty_like(x) if need_dx else None
    dw = torch.empty(C, dtype=torch.float32, device=dev) if need_dw else None
    db = torch.empty(C, dtype=torch.float32, device=dev) if need_db else None
    ws = None
    if need_dw or need_db:
        ws = torch.empty(int(L.lib().miq_gather_layernorm_bwd_workspace(L.i64(R), L.i64(C))), dtype=torch.float32, device=dev)
    w_code = L._DT[weight.dtype] if weight is not None else L.F32
    with L.on_device(dev):
        L.call("miq_gather_layernorm_bwd", L.ptr(gy), L.ptr(x), L.ptr(idx_i32), L.ptr(weight), L.ptr(mean), L.ptr(rstd), L.ptr(dx),
               L.ptr(dw), L.ptr(db), L.ptr(ws), L.i64(R), L.i64(C), L.i32(L.dtype_code(x)), L.i32(w_code), L.i32(L.dtype_code(gy)),
               L.stream_of(x))
    return dx, dw, db


def _ln_backward(ctx_needs, gy, x, idx, w, b, mean, rstd):
    """Shared backward of the gather + LayerNorm functions: gradients for (x, weight, bias)."""
    need_w = w is not None and ctx_needs[1]
    need_b = b is not None and ctx_needs[2]
    if gy.dtype not in (torch.float16, torch.float32):
        gy = gy.float()
    dx, dw, db = gather_layernorm_bwd(gy, x, idx, w, mean, rstd, ctx_needs[0], need_w, need_b)
    if dw is not None:
        dw = dw.to(w.dtype)
    if db is not None:
        db = db.to(b.dtype)
    return dx, dw, db


class GatherLayerNorm(torch.autograd.Function):
    """layer_norm(index_select(x, -1, idx), weight, bias): one kernel forward; backward one row kernel (the scatter
    through the index happens in LDS) plus the column sums of d_weight / d_bias when they are needed."""

    @staticmethod
    def forward(ctx, x, idx_i32, inv_i32, weight, bias, eps, out_dtype):
        x = x.contiguous()
        w = None if weight is None else weight.contiguous()
        b = None if bias is None else bias.contiguous()
        y, mean, rstd = gather_layernorm_fwd(x, idx_i32, w, b, eps, out_dtype)
        ctx.save_for_backward(x, idx_i32, w, b, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, idx, w, b, mean, rstd = ctx.saved_tensors
        needs = (ctx.needs_input_grad[0], ctx.needs_input_grad[3], ctx.needs_input_grad[4])
        dx, dw, db = _ln_backward(needs, gy, x, idx, w, b, mean, rstd)
        return dx, None, None, dw, db, None, None


def masked_softmax_supported(scores, mask) -> bool:
    C = scores.shape[-1]
    ok = scores.is_cuda and scores.dtype in (torch.float16, torch.float32) and scores.dim() == 4
    ok = ok and C % (8 if scores.dtype == torch.float16 else 4) == 0 and C <= 16384
    if mask is not None:
        ok = ok and mask.is_cuda and mask.dtype in (torch.float16, torch.float32) and mask.dim() == 4
        ok = ok and mask.shape[1] == 1 and mask.shape[0] in (1, scores.shape[0]) and tuple(mask.shape[2:]) == tuple(scores.shape[2:])
        ok = ok and not (scores.dtype == torch.float32 and mask.dtype == torch.float16)
    return bool(ok)


class MaskedSoftmax(torch.autograd.Function):
    """softmax(max(scores / div + mask, finfo.min), dim=-1, dtype=float32).to(scores.dtype) for [B, H, Q, K]
    scores and a [B or 1, 1, Q, K] mask (or None), one kernel per direction.  The reference's backward uses the fp32
    probabilities, not the rounded output: the backward rebuilds them from the scores, the mask and the forward's per-row
    maximum and reciprocal sum (miq_masked_softmax_bwd_scores)."""

    @staticmethod
    def forward(ctx, scores, mask, div):
        x = scores.contiguous()
        B, H, Q, K = x.shape
        out = torch.empty_like(x)
        m = None
        if mask is not None:
            m = mask.expand(B, 1, Q, K).contiguous() if mask.shape[0] != B and B > 1 else mask.contiguous()
        dev = L.check_gpu(x, m)
        need = ctx.needs_input_grad[0]
        stats = torch.empty((B * H * Q, 2), dtype=torch.float32, device=x.device) if need else None
        with L.on_device(dev):
            L.call("miq_masked_softmax_fwd", L.ptr(x), L.ptr(m), L.ptr(out), L.ptr(stats), L.i64(B * H * Q), L.i64(K), L.i64(H * Q),
                   L.i64(Q), L.f32(float(div)), L.i32(L.dtype_code(x)), L.i32(L.dtype_code(m) if m is not None else L.F32),
                   L.stream_of(x))
        if need:
            ctx.save_for_backward(x, m, stats)
        ctx.div = float(div)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        x, m, stats = ctx.saved_tensors
        B, H, Q, K = x.shape
        dx = torch.empty_like(x)
        with L.on_device(x.device):
            L.call("miq_masked_softmax_bwd_scores", L.ptr(x), L.ptr(m), L.ptr(stats), L.ptr(g), L.ptr(dx), L.i64(B * H * Q),
                   L.i64(K), L.i64(H * Q), L.i64(Q), L.f32(ctx.div), L.i32(L.dtype_code(x)),
                   L.i32(L.dtype_code(m) if m is not None else L.F32), L.stream_of(x))
        return dx, None, None


def rmsnorm_supported(x, weight, bias) -> bool:
    C = x.shape[-1]
    ok = x.is_cuda and x.dtype in (torch.float16, torch.float32) and C % 4 == 0 and 0 < C <= 16384 and x.numel() > 0
    for t in (weight, bias):
        ok = ok and (t is None or (t.is_cuda and t.dtype in (torch.float16, torch.float32) and t.numel() == C
                                   and t.data_ptr() % 16 == 0))
    return bool(ok and weight is not None and x.data_ptr() % 16 == 0)


class RMSNorm(torch.autograd.Function):
    """(weight * (x * rsqrt(mean(x.float()^2, -1) + eps)) [+ bias]).to(x.dtype) with fp32 weight / bias vectors
    (miq_rmsnorm_fwd / _bwd); the caller converts fp16 vectors with .float() so autograd casts their gradients."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        x = x.contiguous()
        w = weight.contiguous()
        b = None if bias is None else bias.contiguous()
        dev = L.check_gpu(x, w, b)
        C = x.shape[-1]
        R = x.numel() // C
        out = torch.empty_like(x)
        rstd = torch.empty(R, dtype=torch.float32, device=dev)
        with L.on_device(dev):
            L.call("miq_rmsnorm_fwd", L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(out), L.ptr(rstd), L.i64(R), L.i64(C),
                   L.f32(float(eps)), L.i32(L.dtype_code(x)), L.stream_of(x))
        ctx.save_for_backward(x, w, rstd)
        ctx.has_bias = b is not None
        return out

    @staticmethod
    def backward(ctx, gy):
        x, w, rstd = ctx.saved_tensors
        gy = gy.contiguous()
        if gy.dtype != x.dtype:
            gy = gy.to(x.dtype)
        C = x.shape[-1]
        R = x.numel() // C
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        dx = torch.empty_like(x) if need_x else None
        dw = torch.empty(C, dtype=torch.float32, device=x.device) if need_w else None
        db = torch.empty(C, dtype=torch.float32, device=x.device) if need_b else None
        ws = None
        if need_w or need_b:
            ws = torch.empty(int(L.lib().miq_rmsnorm_workspace(L.i64(R), L.i64(C))), dtype=torch.float32, device=x.device)
        with L.on_device(x.device):
            L.call("miq_rmsnorm_bwd", L.ptr(gy), L.ptr(x), L.ptr(w), L.ptr(rstd), L.ptr(dx), L.ptr(dw), L.ptr(db), L.ptr(ws),
                   L.i64(R), L.i64(C), L.i32(L.dtype_code(x)), L.stream_of(x))
        return dx, dw, db, None


def _quant_from_table(y2, n_bits, layout, ws):
    R, C = y2.shape
    q = torch.empty_like(y2)
    vec = torch.empty((5, C), dtype=y2.dtype, device=y2.device)
    L.call("miq_act_quant_from_table", L.ptr(y2), L.i64(layout.cluster_size), L.ptr(layout.slot_of), L.i32(layout.n_clusters),
           L.i64(layout.covered), L.i32(int(n_bits)), L.ptr(ws), L.ptr(q), None, L.ptr(vec), L.i64(R), L.i64(C),
           L.i32(L.dtype_code(y2)), L.stream_of(y2))
    return q, vec


def _stat_table(dev, C, layout):
    nbytes = int(L.lib().miq_act_fresh_workspace(L.i32(1), L.i64(C), L.i32(layout.n_clusters), L.i64(layout.covered)))
    return _fresh_workspace(dev, nbytes)


def _layout_args(layout):
    """Layout arguments of the norm kernels that publish statistics: their packs are 4 channels wide, so equal clusters of
    another granularity (or with channels left over) go through the map like unequal ones."""
    by_arithmetic = layout.cluster_size > 0 and layout.cluster_size % 4 == 0 and layout.covered == layout.C
    return (L.i64(layout.cluster_size), L.ptr(None if by_arithmetic else layout.slot_map()), L.i32(layout.n_clusters),
            L.i64(layout.covered))


class RMSNormActQuant(torch.autograd.Function):
    """act_quant(rmsnorm(x)) for a static quantizer calibrating on fresh statistics, in two launches: the norm kernel
    publishes the minima / maxima of its own output (miq_rmsnorm_fwd_stats), the quantize kernel follows
    (miq_act_quant_from_table).  Same values as RMSNorm.apply followed by ActCalibFQ.apply.  holder receives the
    statistics dictionary; returns the quantized tensor."""

    _calls = 0  # forward evaluations so far (tests use it to see that a block took this path)

    @staticmethod
    def forward(ctx, x, weight, bias, eps, n_bits, layout, holder):
        RMSNormActQuant._calls += 1
        x = x.contiguous()
        w = weight.contiguous()
        b = None if bias is None else bias.contiguous()
        dev = L.check_gpu(x, w, b)
        C = x.shape[-1]
        R = x.numel() // C
        y = torch.empty_like(x)
        rstd = torch.empty(R, dtype=torch.float32, device=dev)
        with L.on_device(dev):
            ws = _stat_table(dev, C, layout)
            L.call("miq_rmsnorm_fwd_stats", L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), L.ptr(rstd), L.i64(R), L.i64(C),
                   L.f32(float(eps)), L.i32(L.dtype_code(x)), *_layout_args(layout), L.ptr(ws), L.stream_of(x))
            q, vec = _quant_from_table(y.reshape(R, C), n_bits, layout, ws)
        holder.update(stats_of(vec))
        ctx.save_for_backward(x, w, rstd, y, vec)
        ctx.has_bias, ctx.n_bits, ctx.layout = b is not None, n_bits, layout
        return q.reshape(x.shape)

    @staticmethod
    def backward(ctx, gq):
        x, w, rstd, y, vec = ctx.saved_tensors
        C = x.shape[-1]
        y2 = y.reshape(-1, C)
        gy = act_fresh_quant_bwd([gq.contiguous().reshape(y2.shape)], [y2], [vec], ctx.n_bits, ctx.layout)[0]
        R = y2.shape[0]
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        dx = torch.empty_like(x) if need_x else None
        dw = torch.empty(C, dtype=torch.float32, device=x.device) if need_w else None
        db = torch.empty(C, dtype=torch.float32, device=x.device) if need_b else None
        ws = None
        if need_w or need_b:
            ws = torch.empty(int(L.lib().miq_rmsnorm_workspace(L.i64(R), L.i64(C))), dtype=torch.float32, device=x.device)
        with L.on_device(x.device):
            L.call("miq_rmsnorm_bwd", L.ptr(gy), L.ptr(x), L.ptr(w), L.ptr(rstd), L.ptr(dx), L.ptr(dw), L.ptr(db), L.ptr(ws),
                   L.i64(R), L.i64(C), L.i32(L.dtype_code(x)), L.stream_of(x))
        return dx, dw, db, None, None, None, None


class GatherLayerNormActQuant(torch.autograd.Function):
    """act_quant(layer_norm(index_select(x, -1, idx))) for a static quantizer calibrating on fresh statistics, in two
    launches (miq_gather_layernorm_fwd_stats, miq_act_quant_from_table).  Same values as GatherLayerNorm.apply followed by
    ActCalibFQ.apply."""

    _calls = 0

    @staticmethod
    def forward(ctx, x, idx_i32, weight, bias, eps, out_dtype, n_bits, layout, holder):
        GatherLayerNormActQuant._calls += 1
        x = x.contiguous()
        w = None if weight is None else weight.contiguous()
        b = None if bias is None else bias.contiguous()
        dev = L.check_gpu(x, idx_i32, w, b)
        C = x.shape[-1]
        R = x.numel() // C
        y = torch.empty_like(x, dtype=out_dtype or x.dtype)
        mean = torch.empty(R, dtype=torch.float32, device=dev)
        rstd = torch.empty(R, dtype=torch.float32, device=dev)
        wb = w if w is not None else b
        w_code = L._DT[wb.dtype] if wb is not None else L.F32
        with L.on_device(dev):
            ws = _stat_table(dev, C, layout)
            L.call("miq_gather_layernorm_fwd_stats", L.ptr(x), L.ptr(idx_i32), L.ptr(w), L.ptr(b), L.ptr(y), L.ptr(mean), L.ptr(rstd),
                   L.i64(R), L.i64(C), L.f32(float(eps)), L.i32(L.dtype_code(x)), L.i32(w_code), L.i32(L.dtype_code(y)),
                   *_layout_args(layout), L.ptr(ws), L.stream_of(x))
            q, vec = _quant_from_table(y.reshape(R, C), n_bits, layout, ws)
        holder.update(stats_of(vec))
        ctx.save_for_backward(x, idx_i32, w, b, mean, rstd, y, vec)
        ctx.n_bits, ctx.layout = n_bits, layout
        return q.reshape(y.shape)

    @staticmethod
    def backward(ctx, gq):
        x, idx, w, b, mean, rstd, y, vec = ctx.saved_tensors
        y2 = y.reshape(-1, y.shape[-1])
        gy = act_fresh_quant_bwd([gq.contiguous().reshape(y2.shape)], [y2], [vec], ctx.n_bits, ctx.layout)[0]
        needs = (ctx.needs_input_grad[0], ctx.needs_input_grad[2], ctx.needs_input_grad[3])
        dx, dw, db = _ln_backward(needs, gy, x, idx, w, b, mean, rstd)
        return dx, None, dw, db, None, None, None, None, None


FUSED_MSE = True  # module-wide switch of mse_loss's kernel path (A/B timing)


def mse_supported(a, b) -> bool:
    return bool(FUSED_MSE and a.is_cuda and b.is_cuda and a.dtype == b.dtype == torch.float16 and a.shape == b.shape
                and a.numel() > 0 and torch.is_autocast_enabled())


class MSELossHalf(torch.autograd.Function):
    """torch.nn.functional.mse_loss(a, b) for fp16 tensors under autocast (which evaluates it in fp32) and for fp32
    tensors: fp32 scalar from one streaming pass over the tensors; gradients bit-identical to ATen's (miq_mse_fwd /
    miq_mse_bwd)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        dev = L.check_gpu(a, b)
        out = torch.empty((), dtype=torch.float32, device=dev)
        ws = torch.empty(int(L.lib().miq_mse_workspace()), dtype=torch.float32, device=dev)
        with L.on_device(dev):
            L.call("miq_mse_fwd", L.ptr(a), L.ptr(b), L.ptr(out), L.ptr(ws), L.i64(a.numel()), L.i32(L.dtype_code(a)), L.stream_of(a))
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        need_a, need_b = ctx.needs_input_grad
        if not (need_a or need_b):
            return None, None
        g = g.to(torch.float32).contiguous()
        da = torch.empty_like(a) if need_a else None
        db = torch.empty_like(b) if need_b else None
        with L.on_device(a.device):
            L.call("miq_mse_bwd", L.ptr(a), L.ptr(b), L.ptr(g), L.ptr(da), L.ptr(db), L.i64(a.numel()), L.i32(L.dtype_code(a)),
                   L.stream_of(a))
        return da, db


def mse_loss(a, b, kernel=None):
    """The calibration loss: MSELoss()(a, b) (reference quantize/omniquant.py:339, 511-513).  ``kernel=True`` takes the
    kernel path for fp32 tensors too (same gradients bit for bit, the loss value in another summation order): a step
    that is replayed from a HIP graph must not contain ATen's multi-block reduction."""
    if mse_supported(a, b) or (kernel and a.is_cuda and b.is_cuda and a.dtype == b.dtype and a.shape == b.shape
                               and a.dtype in (torch.float16, torch.float32) and a.numel() > 0
                               and not (a.dtype == torch.float16 and not torch.is_autocast_enabled())):
        return MSELossHalf.apply(a, b)
    return torch.nn.functional.mse_loss(a, b)


def truncate_inplace_(tensors, threshold: float = 1e-2):
    """x[|x| < threshold] = sign(x) * threshold, in place, for a list of contiguous GPU tensors of one dtype
    (miq_truncate_inplace: one launch per 16 tensors)."""
    import ctypes
    tensors = [t for t in tensors if t.numel()]
    if not tensors:
        return
    dev = L.check_gpu(*tensors)
    code = L.dtype_code(tensors[0])
    if any(t.dtype != tensors[0].dtype for t in tensors):
        raise L.MiqError("truncate_inplace_: tensors of one dtype per call")
    with L.on_device(dev):
        for i in range(0, len(tensors), 16):
            part = tensors[i:i + 16]
            sizes = (ctypes.c_int64 * len(part))(*[t.numel() for t in part])
            L.call("miq_truncate_inplace", _ptr_table(part), sizes, L.i32(len(part)), L.f32(float(threshold)), L.i32(code),
                   L.stream_of(part[0]))


def rotary_supported(x, cos, sin) -> bool:
    if not (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float16, torch.float32) and x.numel() > 0):
        return False
    vec = 8 if x.dtype == torch.float16 else 4
    B, H, T, D = x.shape
    ok = D % (2 * vec) == 0 and x.stride(3) == 1 and all(s % vec == 0 and s >= 0 for s in x.stride()[:3])
    ok = ok and x.data_ptr() % 16 == 0
    for t in (cos, sin):
        ok = ok and t.is_cuda and t.dtype == x.dtype and t.dim() == 4 and t.is_contiguous()
        ok = ok and t.shape[1] == 1 and t.shape[0] in (1, B) and tuple(t.shape[2:]) == (T, D)
    return bool(ok)


def _rotary_call(x, cos, sin, out, backward: bool):
    B, H, T, D = x.shape
    tab_bs = T * D if cos.shape[0] > 1 else 0
    with L.on_device(x.device):
        L.call("miq_rotary", L.ptr(x), L.ptr(cos), L.ptr(sin), L.ptr(out), L.i64(B), L.i64(H), L.i64(T), L.i64(D),
               L.i64(x.stride(0)), L.i64(x.stride(1)), L.i64(x.stride(2)), L.i64(out.stride(0)), L.i64(out.stride(1)),
               L.i64(out.stride(2)), L.i64(tab_bs), L.i32(int(backward)), L.i32(L.dtype_code(x)), L.stream_of(x))


class RotaryEmb(torch.autograd.Function):
    """x * cos + rotate_half(x) * sin for a [B, H, T, D] tensor (any strides with a dense last dimension, e.g. the
    transposed view of a projection output) and [B or 1, 1, T, D] tables; the result is contiguous.  The gradient
    comes back in the [B, T, H, D] memory layout of a projection output (as a transposed view), so the view /
    transpose backward in front of the linear needs no copy."""

    @staticmethod
    def forward(ctx, x, cos, sin):
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
        _rotary_call(x, cos, sin, out, False)
        ctx.save_for_backward(cos, sin)
        return out

    @staticmethod
    def backward(ctx, g):
        cos, sin = ctx.saved_tensors
        if not rotary_supported(g, cos, sin):
            g = g.contiguous()
        B, H, T, D = g.shape
        dx = torch.empty((B, T, H, D), dtype=g.dtype, device=g.device).transpose(1, 2)
        _rotary_call(g, cos, sin, dx, True)
        return dx, None, None


def col_sum(x2d, round_to=None):
    """fp32 [C] column sums of a contiguous [R, C] fp32 / fp16 matrix (miq_col_sum); ``round_to=torch.float16``
    rounds every sum to fp16 first (the value an fp16 reduction returns, inf on overflow)."""
    dev = L.check_gpu(x2d)
    R, C = x2d.shape
    out = torch.empty(C, dtype=torch.float32, device=dev)
    n = int(L.lib().miq_col_sum_workspace(L.i64(R), L.i64(C)))
    ws = torch.empty(n, dtype=torch.float32, device=dev) if n > C else None
    with L.on_device(dev):
        L.call("miq_col_sum", L.ptr(x2d), L.ptr(out), L.ptr(ws), L.i64(R), L.i64(C), L.i32(L.dtype_code(x2d)),
               L.i32(L.F16 if round_to == torch.float16 else L.F32), L.stream_of(x2d))
    return out


def linear_supported(x, weight, bias) -> bool:
    ok = bias is not None and x.is_cuda and weight.is_cuda and bias.is_cuda and weight.dim() == 2 and x.dim() >= 2
    ok = ok and all(t.dtype in (torch.float16, torch.float32) for t in (x, weight, bias))
    ok = ok and (torch.is_autocast_enabled() or x.dtype == weight.dtype == bias.dtype) and x.numel() > 0
    return bool(ok)


class LinearBias(torch.autograd.Function):
    """torch.nn.functional.linear(x, weight, bias) with the same GEMMs forward and backward; only the bias gradient
    differs: one streaming column-sum kernel (miq_col_sum) instead of ATen's reduction over the slow dimension.
    Under autocast the operands are cast to the autocast dtype first, like the autocast rule of linear."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.in_dtypes = (x.dtype, weight.dtype, bias.dtype)
        if torch.is_autocast_enabled():
            dt = torch.get_autocast_dtype("cuda")
            x, weight, bias = x.to(dt), weight.to(dt), bias.to(dt)
        with torch.autocast("cuda", enabled=False):
            y = torch.nn.functional.linear(x, weight, bias)
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        need_x, need_w, need_b = ctx.needs_input_grad
        g2 = g.reshape(-1, w.shape[0])
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        dx = dw = db = None
        with torch.autocast("cuda", enabled=False):
            if need_x:
                dx = g2.mm(w).reshape(x.shape).to(ctx.in_dtypes[0])
            if need_w:
                dw = g2.t().mm(x.reshape(-1, x.shape[-1])).to(ctx.in_dtypes[1])
            if need_b:
                # rounded to the compute dtype first, like the reduction it replaces: an fp16 sum that overflows must
                # become inf so that the loss scaler skips the step exactly when it would with F.linear
                db = col_sum(g2, round_to=g2.dtype).to(ctx.in_dtypes[2])
        return dx, dw, db


class ActEvalFQ(torch.autograd.Function):
    """Frozen per-channel scales: out = fq(x); dx = g * [level inside the grid]."""

    @staticmethod
    def forward(ctx, x, scale_vec, rzp_vec, n_bits):
        x = x.contiguous()
        x2 = x.reshape(-1, x.shape[-1])
        out, _ = colquant_fwd(x2, scale_vec, rzp_vec, n_bits)
        ctx.save_for_backward(x2, scale_vec, rzp_vec)
        ctx.n_bits = n_bits
        return out.reshape(x.shape)

    @staticmethod
    def backward(ctx, g):
        x2, s, z = ctx.saved_tensors
        dx = colquant_bwd(g.contiguous().reshape(x2.shape), x2, s, z, ctx.n_bits)
        return dx.reshape(g.shape), None, None, None


TENSOR_FOLD = True  # module-wide switch of _tensor_fold (A/B timing)


def _tensor_fold(R: int, C: int, layout) -> int:
    """Per-tensor statistics do not depend on how the elements are cut into rows: a narrow [R, C] slab (the
    [.., head_dim] operands of the attention matmuls) is handed to the column kernels as [R / k, C * k], so that
    every lane of their 64-lane x 16-byte tiles has a channel pack to work on (C = 128 fp16 fills 16 of 64 lanes:
    measured 66 us instead of 32 us for 8.4 M elements).  Returns k (1 = leave as is)."""
    if not TENSOR_FOLD or layout.n_clusters != 1 or layout.covered != C or C >= 2048 or C % 8:
        return 1
    k = 1
    while C * k * 2 <= 4096 and R % (k * 2) == 0:
        k *= 2
    return k


class ActCalibFQ(torch.autograd.Function):
    """Calibration-mode forward with freshly computed statistics (no cached values): the scale is a
    function of x, so the backward routes the scale gradient to the extreme elements."""

    @staticmethod
    def forward(ctx, x, n_bits, metric, layout, cached_min, cached_max, holder, dp=None):
        x = x.contiguous()
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        synced = dp is not None and dp.active and dp.sync_stats
        # (with statistics synchronised across ranks the vectors that are all-reduced keep the caller's width, whatever
        # the local row count)
        k = _tensor_fold(x2.shape[0], C, layout) if (x2.numel() and not synced) else 1
        if k > 1:
            x2 = x2.reshape(-1, C * k)
            layout = ClusterLayout.get(None, C * k, x.device)
            if cached_min is not None:  # one cluster: all entries are equal
                cached_min = cached_min[:1].expand(C * k).contiguous()
                cached_max = cached_max[:1].expand(C * k).contiguous()
        ctx.dp = dp
        ctx.fresh = cached_min is None
        ctx.layout = layout
        ctx.n_bits = n_bits
        ctx.vec_path = ctx.fresh and not synced and x2.numel() > 0
        if ctx.vec_path:  # the training-step case: two launches, statistics merged by atomics
            outs, _, vecs = act_fresh_quant_fwd([x2], n_bits, layout)
            out, st = outs[0], stats_of(vecs[0])
            ctx.save_for_backward(x2, vecs[0])
        else:
            if synced:
                st = act_calibrate(x2, n_bits, metric, layout, cached_min, cached_max, dp=dp)
                out, _ = colquant_fwd(x2, st["scale"], st["rzp"], n_bits)
            else:
                out, _, st = act_calib_quant_fwd(x2, n_bits, metric, layout, cached_min, cached_max)
            ctx.save_for_backward(x2, st["scale"], st["rzp"], st["chan_min"], st["chan_max"], st["cmin"], st["cmax"])
        if k > 1:  # publish vectors of the caller's channel count
            st = {n: (v if n.startswith("chan_") or n == "vec" else v[:C]) for n, v in st.items()}
        holder.update(st)
        return out.reshape(x.shape)

    @staticmethod
    def backward(ctx, g):
        if not ctx.fresh:
            raise L.MiqError("backward through cached activation statistics is not defined: call free() on the "
                             "quantizer after every optimizer step (the calibration loop does)")
        if ctx.vec_path:
            x2, vec = ctx.saved_tensors
            dx = act_fresh_quant_bwd([g.contiguous().reshape(x2.shape)], [x2], [vec], ctx.n_bits, ctx.layout)[0]
            return dx.reshape(g.shape), None, None, None, None, None, None, None
        x2, s, z, chan_min, chan_max, kmin, kmax = ctx.saved_tensors
        dx = colquant_bwd(g.contiguous().reshape(x2.shape), x2, s, z, ctx.n_bits,
                          route=(chan_min, chan_max, kmin, kmax, ctx.layout), dp=ctx.dp)
        return dx.reshape(g.shape), None, None, None, None, None, None, None


class ActCalibFQBatch(torch.autograd.Function):
    """apply(n_bits, layout, holders, x_0, x_1, ...) -> (out_0, out_1, ...): several activations of one shape, dtype
    and cluster layout (the q / k / v projection outputs of a block, say) calibrated with fresh statistics and
    quantized by one pair of launches, and one backward; holders[i] receives the statistics of x_i."""

    @staticmethod
    def forward(ctx, n_bits, layout, holders, *xs):
        C = xs[0].shape[-1]
        x2s = [x.contiguous().reshape(-1, C) for x in xs]
        k = _tensor_fold(x2s[0].shape[0], C, layout)  # narrow per-tensor slabs as wide rows, as in ActCalibFQ
        if k > 1:
            x2s = [x.reshape(-1, C * k) for x in x2s]
            layout = ClusterLayout.get(None, C * k, xs[0].device)
        outs, _, vecs = act_fresh_quant_fwd(x2s, n_bits, layout)
        for h, v in zip(holders, vecs):
            st = stats_of(v)
            if k > 1:
                st = {n: (t if n == "vec" else t[:C]) for n, t in st.items()}
            h.update(st)
        ctx.layout, ctx.n_bits, ctx.n = layout, n_bits, len(xs)
        ctx.save_for_backward(*x2s, *vecs)
        ctx.set_materialize_grads(False)
        return tuple(o.reshape(x.shape) for o, x in zip(outs, xs))

    @staticmethod
    def backward(ctx, *gs):
        n = ctx.n
        x2s, vecs = list(ctx.saved_tensors[:n]), list(ctx.saved_tensors[n:])
        live = [i for i in range(n) if gs[i] is not None and ctx.needs_input_grad[3 + i]]
        dxs = [None] * n
        if live:
            g2s = [gs[i].contiguous().reshape(x2s[i].shape) for i in live]
            res = act_fresh_quant_bwd(g2s, [x2s[i] for i in live], [vecs[i] for i in live], ctx.n_bits, ctx.layout)
            for i, d in zip(live, res):
                dxs[i] = d.reshape(gs[i].shape)
        return (None, None, None, *dxs)


# ---------------------------------------------------------------------------------------------
# LET-fused weight quantizer
# ---------------------------------------------------------------------------------------------
def let_supported(w, group=0) -> bool:
    return (w.is_cuda and w.dtype == torch.float32 and w.dim() == 2 and not group and w.shape[1] % 4 == 0
            and w.shape[1] <= 16384 and w.is_contiguous())


def let_rowquant_fwd(w, cs, rs, rs_div, shift, up, low, n_bits, symmetric, out_dtype=None):
    dev = L.check_gpu(w, cs, rs, shift, up, low)
    O, I = w.shape
    out = torch.empty_like(w, dtype=out_dtype or w.dtype)
    scale = torch.empty((O, 1), dtype=w.dtype, device=dev)
    rzp = torch.empty_like(scale)
    wshift = torch.empty(O, dtype=w.dtype, device=dev) if shift is not None else None
    with L.on_device(dev):
        L.call("miq_let_rowquant_fwd", L.ptr(w), L.ptr(cs), L.ptr(rs), L.i32(int(bool(rs_div))), L.ptr(shift), L.ptr(up),
               L.ptr(low), L.ptr(out), L.ptr(scale), L.ptr(rzp), L.ptr(wshift), L.i64(O), L.i64(I), L.i32(int(n_bits)),
               L.i32(int(bool(symmetric))), L.i32(_flags(False)), L.i32(L.dtype_code(out)), L.stream_of(w))
    return out, wshift, scale, rzp


def let_rowquant_bwd(g, g_wshift, w, cs, rs, rs_div, shift, up, low, n_bits, symmetric, want_cs=True, want_rs=True,
                     want_shift=True):
    dev = L.check_gpu(g, g_wshift, w, cs, rs, shift, up, low)
    O, I = w.shape
    want_cs = want_cs and cs is not None
    want_rs = want_rs and rs is not None
    want_shift = want_shift and shift is not None and g_wshift is not None
    d_cs = torch.empty_like(cs) if want_cs else None
    d_rs = torch.empty_like(rs) if want_rs else None
    d_shift = torch.empty_like(shift) if want_shift else None
    d_up = torch.empty_like(up) if up is not None else None
    d_low = torch.empty_like(low) if low is not None else None
    ws = None
    if want_cs or want_shift:
        ws = torch.empty(int(L.lib().miq_let_workspace(L.i64(O), L.i64(I))), dtype=torch.float32, device=dev)
    with L.on_device(dev):
        L.call("miq_let_rowquant_bwd", L.ptr(g), L.ptr(g_wshift), L.ptr(w), L.ptr(cs), L.ptr(rs), L.i32(int(bool(rs_div))),
               L.ptr(shift), L.ptr(up), L.ptr(low), L.ptr(d_cs), L.ptr(d_rs), L.ptr(d_shift), L.ptr(d_up), L.ptr(d_low),
               L.ptr(ws), L.i64(O), L.i64(I), L.i32(int(n_bits)), L.i32(int(bool(symmetric))),
               L.i32(_flags(False)), L.i32(L.dtype_code(g)), L.stream_of(w))
    return d_cs, d_rs, d_shift, d_up, d_low


class LetRowFQ(torch.autograd.Function):
    """(wq, wshift, scale, rzp) = quantize((w * cs) / rs or * rs), wshift = w @ shift; w is a constant.
    ``half_out``: store wq in fp16 (for consumers that would cast it anyway, i.e. linears under autocast);
    the incoming gradient is then fp16 too and is read as such."""

    @staticmethod
    def forward(ctx, w, cs, rs, rs_div, shift, up, low, n_bits, symmetric, half_out=False):
        flat = lambda t: None if t is None else t.detach().reshape(-1).contiguous()
        csf, rsf, shf = flat(cs), flat(rs), flat(shift)
        out, wshift, scale, rzp = let_rowquant_fwd(w, csf, rsf, rs_div, shf, up, low, n_bits, symmetric,
                                                   out_dtype=torch.float16 if half_out else None)
        ctx.save_for_backward(w, csf, rsf, shf, up, low)
        ctx.cfg = (bool(rs_div), int(n_bits), bool(symmetric))
        ctx.half_out = bool(half_out)
        ctx.shapes = tuple(None if t is None else t.shape for t in (cs, rs, shift))
        ctx.mark_non_differentiable(scale, rzp)
        if wshift is None:
            wshift = w.new_zeros(())  # placeholder so that the output arity is fixed
            ctx.mark_non_differentiable(wshift)
        ctx.set_materialize_grads(False)  # no zero-filled gradients for the scale / zero-point / placeholder outputs
        return out, wshift, scale, rzp

    @staticmethod
    def backward(ctx, g, g_wshift, _gs, _gz):
        w, cs, rs, sh, up, low = ctx.saved_tensors
        rs_div, n_bits, symmetric = ctx.cfg
        need = ctx.needs_input_grad
        if (g is None and g_wshift is None) or not any(need[i] for i in (1, 2, 4, 5, 6)):
            return (None,) * 10  # nothing was used downstream, or nothing trainable went in
        if g is None:  # only w @ shift was used
            g = torch.zeros(w.shape, dtype=torch.float16 if ctx.half_out else w.dtype, device=w.device)
        if g_wshift is None and sh is not None:
            g_wshift = torch.zeros(w.shape[0], dtype=w.dtype, device=w.device)
        gw = g_wshift.contiguous() if (sh is not None and g_wshift is not None) else None
        d_cs, d_rs, d_shift, d_up, d_low = let_rowquant_bwd(g.contiguous(), gw, w, cs, rs, rs_div, sh, up, low, n_bits,
                                                            symmetric, want_cs=need[1], want_rs=need[2], want_shift=need[4])
        shp = ctx.shapes
        rs_ = lambda t, s: None if t is None else t.reshape(s)
        return None, rs_(d_cs, shp[0]), rs_(d_rs, shp[1]), None, rs_(d_shift, shp[2]), d_up, d_low, None, None, None


# ---------------------------------------------------------------------------------------------
# exponential / mixed-bit row-wise quantizers
# ---------------------------------------------------------------------------------------------
def _check_lwc(x, up, low, rows):
    if up is not None and (up.numel() != rows or low.numel() != rows or up.dtype != x.dtype or low.dtype != x.dtype):
        raise L.MiqError(f"LWC factors must be {rows} values of dtype {x.dtype}")


def rowquant_exp2_fwd(x, n_bits: int, up=None, low=None, group: int = 0):
    """Returns (out, scale[rows,1])."""
    dev = L.check_gpu(x, up, low)
    O, I, g, rows = _rows_of(x, group, up is not None)
    _check_lwc(x, up, low, rows)
    out = torch.empty_like(x)
    scale = torch.empty((rows, 1), dtype=x.dtype, device=dev)
    with L.on_device(dev):
        L.call("miq_rowquant_exp2_fwd", L.ptr(x), L.ptr(up), L.ptr(low), L.ptr(out), L.ptr(scale), L.i64(O), L.i64(I),
               L.i64(g), L.i32(int(n_bits)), L.i32(_flags(False)), L.i32(L.dtype_code(x)), L.stream_of(x))
    return out, scale


def rowquant_exp2_bwd(grad_out, x, n_bits: int, up=None, low=None, group: int = 0, want_dx=True):
    dev = L.check_gpu(grad_out, x, up, low)
    O, I, g, rows = _rows_of(x, group, up is not None)
    dx = torch.empty_like(x) if want_dx else None
    d_up = torch.empty_like(up) if up is not None else None
    d_low = torch.empty_like(low) if low is not None else None
    with L.on_device(dev):
        L.call("miq_rowquant_exp2_bwd", L.ptr(grad_out), L.ptr(x), L.ptr(up), L.ptr(low), L.ptr(dx), L.ptr(d_up),
               L.ptr(d_low), L.i64(O), L.i64(I), L.i64(g), L.i32(int(n_bits)), L.i32(_flags(False)),
               L.i32(L.dtype_code(x)), L.stream_of(x))
    return dx, d_up, d_low


def rowquant_exp2_fwd_batch(xs, n_bits: int, ups=None, lows=None):
    """The exponent (v2) quantizer over several [O_i, I] matrices of one width and dtype in one launch.
    Returns (outs, scales) lists; bit-identical to rowquant_exp2_fwd per matrix."""
    import ctypes
    lwc = ups is not None
    dev = L.check_gpu(*xs, *(ups or []), *(lows or []))
    _check_batch(xs, ups, lows)
    I = xs[0].shape[1]
    outs = [torch.empty_like(x) for x in xs]
    scales = [torch.empty((x.shape[0], 1), dtype=x.dtype, device=dev) for x in xs]
    O = (ctypes.c_int64 * len(xs))(*[x.shape[0] for x in xs])
    with L.on_device(dev):
        L.call("miq_rowquant_exp2_fwd_batch", L.i32(len(xs)), _ptr_table(xs), _ptr_table(ups) if lwc else None,
               _ptr_table(lows) if lwc else None, _ptr_table(outs), _ptr_table(scales), O, L.i64(I), L.i32(int(n_bits)),
               L.i32(_flags(False)), L.i32(L.dtype_code(xs[0])), L.stream_of(xs[0]))
    return outs, scales


def rowquant_exp2_bwd_batch(gs, xs, n_bits: int, ups=None, lows=None, want_dx: bool = True):
    """Backward of the above.  Returns (dxs|None, d_ups|None, d_lows|None) lists."""
    import ctypes
    lwc = ups is not None
    dev = L.check_gpu(*gs, *xs, *(ups or []), *(lows or []))
    _check_batch(xs, ups, lows, gs)
    I = xs[0].shape[1]
    dxs = [torch.empty_like(x) for x in xs] if want_dx else None
    d_ups = [torch.empty_like(u) for u in ups] if lwc else None
    d_lows = [torch.empty_like(u) for u in lows] if lwc else None
    O = (ctypes.c_int64 * len(xs))(*[x.shape[0] for x in xs])
    with L.on_device(dev):
        L.call("miq_rowquant_exp2_bwd_batch", L.i32(len(xs)), _ptr_table(gs), _ptr_table(xs), _ptr_table(ups) if lwc else None,
               _ptr_table(lows) if lwc else None, _ptr_table(dxs) if want_dx else None, _ptr_table(d_ups) if lwc else None,
               _ptr_table(d_lows) if lwc else None, O, L.i64(I), L.i32(int(n_bits)), L.i32(_flags(False)),
               L.i32(L.dtype_code(xs[0])), L.stream_of(xs[0]))
    return dxs, d_ups, d_lows


def rowquant_exp1_fwd(x, n_bits: int, symmetric: bool, up=None, low=None, group: int = 0):
    """Returns (out, scale[rows,1], mid[rows,1]|None)."""
    dev = L.check_gpu(x, up, low)
    O, I, g, rows = _rows_of(x, group, up is not None)
    _check_lwc(x, up, low, rows)
    out = torch.empty_like(x)
    scale = torch.empty((rows, 1), dtype=x.dtype, device=dev)
    mid = None if symmetric else torch.empty((rows, 1), dtype=x.dtype, device=dev)
    with L.on_device(dev):
        L.call("miq_rowquant_exp1_fwd", L.ptr(x), L.ptr(up), L.ptr(low), L.ptr(out), L.ptr(scale), L.ptr(mid), L.i64(O),
               L.i64(I), L.i64(g), L.i32(int(n_bits)), L.i32(int(bool(symmetric))), L.i32(_flags(False)),
               L.i32(L.dtype_code(x)), L.stream_of(x))
    return out, scale, mid


def rowquant_exp1_bwd(grad_out, x, n_bits: int, symmetric: bool, up=None, low=None, group: int = 0, want_dx=True):
    dev = L.check_gpu(grad_out, x, up, low)
    O, I, g, rows = _rows_of(x, group, up is not None)
    dx = torch.empty_like(x) if want_dx else None
    d_up = torch.empty_like(up) if up is not None else None
    d_low = torch.empty_like(low) if low is not None else None
    with L.on_device(dev):
        L.call("miq_rowquant_exp1_bwd", L.ptr(grad_out), L.ptr(x), L.ptr(up), L.ptr(low), L.ptr(dx), L.ptr(d_up),
               L.ptr(d_low), L.i64(O), L.i64(I), L.i64(g), L.i32(int(n_bits)), L.i32(int(bool(symmetric))),
               L.i32(_flags(False)), L.i32(L.dtype_code(x)), L.stream_of(x))
    return dx, d_up, d_low


def rowquant_exp1_fwd_batch(xs, n_bits: int, symmetric: bool, ups=None, lows=None):
    """The exponent (v1) quantizer over several [O_i, I] matrices of one width and dtype in one launch.
    Returns (outs, scales, mids|None) lists; bit-identical to rowquant_exp1_fwd per matrix."""
    import ctypes
    lwc = ups is not None
    dev = L.check_gpu(*xs, *(ups or []), *(lows or []))
    _check_batch(xs, ups, lows)
    I = xs[0].shape[1]
    outs = [torch.empty_like(x) for x in xs]
    scales = [torch.empty((x.shape[0], 1), dtype=x.dtype, device=dev) for x in xs]
    mids = None if symmetric else [torch.empty((x.shape[0], 1), dtype=x.dtype, device=dev) for x in xs]
    O = (ctypes.c_int64 * len(xs))(*[x.shape[0] for x in xs])
    with L.on_device(dev):
        L.call("miq_rowquant_exp1_fwd_batch", L.i32(len(xs)), _ptr_table(xs), _ptr_table(ups) if lwc else None,
               _ptr_table(lows) if lwc else None, _ptr_table(outs), _ptr_table(scales), _ptr_table(mids) if mids else None, O,
               L.i64(I), L.i32(int(n_bits)), L.i32(int(bool(symmetric))), L.i32(_flags(False)), L.i32(L.dtype_code(xs[0])),
               L.stream_of(xs[0]))
    return outs, scales, mids


def rowquant_exp1_bwd_batch(gs, xs, n_bits: int, symmetric: bool, ups=None, lows=None, want_dx: bool = True):
    """Backward of the above.  Returns (dxs|None, d_ups|None, d_lows|None) lists."""
    import ctypes
    lwc = ups is not None
    dev = L.check_gpu(*gs, *xs, *(ups or []), *(lows or []))
    _check_batch(xs, ups, lows, gs)
    I = xs[0].shape[1]
    dxs = [torch.empty_like(x) for x in xs] if want_dx else None
    d_ups = [torch.empty_like(u) for u in ups] if lwc else None
    d_lows = [torch.empty_like(u) for u in lows] if lwc else None
    O = (ctypes.c_int64 * len(xs))(*[x.shape[0] for x in xs])
    with L.on_device(dev):
        L.call("miq_rowquant_exp1_bwd_batch", L.i32(len(xs)), _ptr_table(gs), _ptr_table(xs), _ptr_table(ups) if lwc else None,
               _ptr_table(lows) if lwc else None, _ptr_table(dxs) if want_dx else None, _ptr_table(d_ups) if lwc else None,
               _ptr_table(d_lows) if lwc else None, O, L.i64(I), L.i32(int(n_bits)), L.i32(int(bool(symmetric))),
               L.i32(_flags(False)), L.i32(L.dtype_code(xs[0])), L.stream_of(xs[0]))
    return dxs, d_ups, d_lows


def rowquant_mix_fwd(x, n_bits_rows, symmetric: bool, up=None, low=None, group: int = 0, hard=False, threshold=0.0):
    dev = L.check_gpu(x, up, low, n_bits_rows)
    if x.dtype != torch.float32 or n_bits_rows.dtype != torch.float32:
        raise L.MiqError("the learnable mixed-bit quantizer is float32 only")
    O, I, g, rows = _rows_of(x, group, up is not None)
    _check_lwc(x, up, low, rows)
    if n_bits_rows.numel() != rows:
        raise L.MiqError(f"n_bits must have {rows} entries")
    out = torch.empty_like(x)
    with L.on_device(dev):
        L.call("miq_rowquant_mix_fwd", L.ptr(x), L.ptr(up), L.ptr(low), L.ptr(n_bits_rows), L.ptr(out), L.i64(O), L.i64(I),
               L.i64(g), L.i32(int(bool(symmetric))), L.i32(int(bool(hard))), L.f32(float(threshold)),
               L.i32(_flags(False)), L.i32(L.F32), L.stream_of(x))
    return out


def rowquant_mix_bwd(grad_out, x, n_bits_rows, symmetric: bool, up=None, low=None, group: int = 0, hard=False,
                     threshold=0.0, want_dx=True):
    dev = L.check_gpu(grad_out, x, up, low, n_bits_rows)
    O, I, g, rows = _rows_of(x, group, up is not None)
    dx = torch.empty_like(x) if want_dx else None
    d_up = torch.empty_like(up) if up is not None else None
    d_low = torch.empty_like(low) if low is not None else None
    d_n = torch.empty_like(n_bits_rows)
    with L.on_device(dev):
        L.call("miq_rowquant_mix_bwd", L.ptr(grad_out), L.ptr(x), L.ptr(up), L.ptr(low), L.ptr(n_bits_rows), L.ptr(dx),
               L.ptr(d_up), L.ptr(d_low), L.ptr(d_n), L.i64(O), L.i64(I), L.i64(g), L.i32(int(bool(symmetric))),
               L.i32(int(bool(hard))), L.f32(float(threshold)), L.i32(_flags(False)), L.i32(L.F32),
               L.stream_of(x))
    return dx, d_up, d_low, d_n


def _check_mix_batch(xs, bits):
    if len(bits) != len(xs):
        raise L.MiqError(f"{len(bits)} bit-width vectors for {len(xs)} matrices")
    for x, nb in zip(xs, bits):
        if x.dtype != torch.float32 or nb.dtype != torch.float32:
            raise L.MiqError("the learnable mixed-bit quantizer is float32 only")
        if nb.numel() != x.shape[0] or not nb.is_contiguous():
            raise L.MiqError(f"every n_bits vector must be a contiguous tensor of {x.shape[0]} entries")


def rowquant_mix_fwd_batch(xs, bits, symmetric: bool, ups=None, lows=None, hard=False, threshold=0.0):
    """The learnable mixed-bit quantizer over several [O_i, I] fp32 matrices of one width in one launch.
    Returns the list of outputs; bit-identical to rowquant_mix_fwd per matrix."""
    import ctypes
    lwc = ups is not None
    dev = L.check_gpu(*xs, *bits, *(ups or []), *(lows or []))
    _check_batch(xs, ups, lows)
    _check_mix_batch(xs, bits)
    outs = [torch.empty_like(x) for x in xs]
    O = (ctypes.c_int64 * len(xs))(*[x.shape[0] for x in xs])
    with L.on_device(dev):
        L.call("miq_rowquant_mix_fwd_batch", L.i32(len(xs)), _ptr_table(xs), _ptr_table(ups) if lwc else None,
               _ptr_table(lows) if lwc else None, _ptr_table(bits), _ptr_table(outs), O, L.i64(xs[0].shape[1]),
               L.i32(int(bool(symmetric))), L.i32(int(bool(hard))), L.f32(float(threshold)), L.i32(_flags(False)),
               L.i32(L.F32), L.stream_of(xs[0]))
    return outs


def rowquant_mix_bwd_batch(gs, xs, bits, symmetric: bool, ups=None, lows=None, hard=False, threshold=0.0, want_dx=True):
    """Backward of the above.  Returns (dxs|None, d_ups|None, d_lows|None, d_bits) lists."""
    import ctypes
    lwc = ups is not None
    dev = L.check_gpu(*gs, *xs, *bits, *(ups or []), *(lows or []))
    _check_batch(xs, ups, lows, gs)
    _check_mix_batch(xs, bits)
    dxs = [torch.empty_like(x) for x in xs] if want_dx else None
    d_ups = [torch.empty_like(u) for u in ups] if lwc else None
    d_lows = [torch.empty_like(u) for u in lows] if lwc else None
    d_bits = [torch.empty_like(b) for b in bits]
    O = (ctypes.c_int64 * len(xs))(*[x.shape[0] for x in xs])
    with L.on_device(dev):
        L.call("miq_rowquant_mix_bwd_batch", L.i32(len(xs)), _ptr_table(gs), _ptr_table(xs), _ptr_table(ups) if lwc else None,
               _ptr_table(lows) if lwc else None, _ptr_table(bits), _ptr_table(dxs) if want_dx else None,
               _ptr_table(d_ups) if lwc else None, _ptr_table(d_lows) if lwc else None, _ptr_table(d_bits), O,
               L.i64(xs[0].shape[1]), L.i32(int(bool(symmetric))), L.i32(int(bool(hard))), L.f32(float(threshold)),
               L.i32(_flags(False)), L.i32(L.F32), L.stream_of(xs[0]))
    return dxs, d_ups, d_lows, d_bits


class Exp2RowFQ(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, up, low, n_bits, group):
        x = x.contiguous()
        out, scale = rowquant_exp2_fwd(x, n_bits, up, low, group)
        ctx.save_for_backward(x, up, low)
        ctx.cfg = (n_bits, group)
        ctx.mark_non_differentiable(scale)
        ctx.set_materialize_grads(False)
        return out, scale

    @staticmethod
    def backward(ctx, g, _gs):
        x, up, low = ctx.saved_tensors
        n_bits, group = ctx.cfg
        need_dx = ctx.needs_input_grad[0]
        if g is None or not (need_dx or (up is not None and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]))):
            return (None,) * 5
        dx, d_up, d_low = rowquant_exp2_bwd(g.contiguous(), x, n_bits, up, low, group, want_dx=need_dx)
        return dx, d_up, d_low, None, None


class Exp2RowFQBatch(torch.autograd.Function):
    """apply(n_bits, lwc, x_0, [up_0, low_0,] x_1, ...) -> (out_0, scale_0, out_1, ...): several weight matrices of one
    width through one forward and one backward launch of the exponent (v2) quantizer."""

    @staticmethod
    def forward(ctx, n_bits, lwc, *ts):
        step = 3 if lwc else 1
        xs = [t.contiguous() for t in ts[0::step]]
        ups = list(ts[1::step]) if lwc else None
        lows = list(ts[2::step]) if lwc else None
        outs, scales = rowquant_exp2_fwd_batch(xs, n_bits, ups, lows)
        ctx.cfg = (n_bits, lwc, len(xs))
        ctx.save_for_backward(*xs, *(ups or []), *(lows or []))
        ret = []
        for o, s in zip(outs, scales):
            ret += [o, s]
        ctx.mark_non_differentiable(*scales)
        ctx.set_materialize_grads(False)
        return tuple(ret)

    @staticmethod
    def backward(ctx, *grads):
        n_bits, lwc, n = ctx.cfg
        saved = ctx.saved_tensors
        xs = list(saved[:n])
        ups = list(saved[n:2 * n]) if lwc else None
        lows = list(saved[2 * n:3 * n]) if lwc else None
        gs = [(g if g is not None else torch.zeros_like(x)).contiguous() for g, x in zip(grads[0::2], xs)]
        step = 3 if lwc else 1
        need_dx = any(ctx.needs_input_grad[2 + i * step] for i in range(n))
        dxs, d_ups, d_lows = rowquant_exp2_bwd_batch(gs, xs, n_bits, ups, lows, want_dx=need_dx)
        out = [None, None]
        for i in range(n):
            out.append(dxs[i] if need_dx else None)
            if lwc:
                out += [d_ups[i], d_lows[i]]
        return tuple(out)


class Exp1RowFQ(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, up, low, n_bits, symmetric, group):
        x = x.contiguous()
        out, scale, mid = rowquant_exp1_fwd(x, n_bits, symmetric, up, low, group)
        ctx.save_for_backward(x, up, low)
        ctx.cfg = (n_bits, symmetric, group)
        if mid is None:
            mid = scale.new_empty(0)
        ctx.mark_non_differentiable(scale, mid)
        ctx.set_materialize_grads(False)
        return out, scale, mid

    @staticmethod
    def backward(ctx, g, _gs, _gm):
        x, up, low = ctx.saved_tensors
        n_bits, symmetric, group = ctx.cfg
        need_dx = ctx.needs_input_grad[0]
        if g is None or not (need_dx or (up is not None and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]))):
            return (None,) * 6
        dx, d_up, d_low = rowquant_exp1_bwd(g.contiguous(), x, n_bits, symmetric, up, low, group, want_dx=need_dx)
        return dx, d_up, d_low, None, None, None


class MixRowFQ(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n_bits_param, up, low, symmetric, group, hard, threshold):
        x = x.contiguous()
        nb = n_bits_param.detach().reshape(-1).contiguous()
        out = rowquant_mix_fwd(x, nb, symmetric, up, low, group, hard, threshold)
        ctx.save_for_backward(x, nb, up, low)
        ctx.cfg = (symmetric, group, hard, threshold, n_bits_param.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        x, nb, up, low = ctx.saved_tensors
        symmetric, group, hard, threshold, nshape = ctx.cfg
        dx, d_up, d_low, d_n = rowquant_mix_bwd(g.contiguous(), x, nb, symmetric, up, low, group, hard, threshold,
                                                want_dx=ctx.needs_input_grad[0])
        return dx, d_n.reshape(nshape), d_up, d_low, None, None, None, None


class MixRowFQBatch(torch.autograd.Function):
    """apply(symmetric, hard, threshold, lwc, x_0, n_bits_0, [up_0, low_0,] x_1, ...) -> (out_0, out_1, ...): several
    fp32 weight matrices of one width through one forward and one backward launch of the learnable mixed-bit quantizer."""

    @staticmethod
    def forward(ctx, symmetric, hard, threshold, lwc, *ts):
        step = 4 if lwc else 2
        xs = [t.contiguous() for t in ts[0::step]]
        params = ts[1::step]
        bits = [p.detach().reshape(-1).contiguous() for p in params]
        ups = list(ts[2::step]) if lwc else None
        lows = list(ts[3::step]) if lwc else None
        outs = rowquant_mix_fwd_batch(xs, bits, symmetric, ups, lows, hard, threshold)
        ctx.cfg = (symmetric, hard, threshold, lwc, len(xs), [p.shape for p in params])
        ctx.save_for_backward(*xs, *bits, *(ups or []), *(lows or []))
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        symmetric, hard, threshold, lwc, n, nshapes = ctx.cfg
        saved = ctx.saved_tensors
        xs, bits = list(saved[:n]), list(saved[n:2 * n])
        ups = list(saved[2 * n:3 * n]) if lwc else None
        lows = list(saved[3 * n:4 * n]) if lwc else None
        gs = [(g if g is not None else torch.zeros_like(x)).contiguous() for g, x in zip(grads, xs)]
        step = 4 if lwc else 2
        need_dx = any(ctx.needs_input_grad[4 + i * step] for i in range(n))
        dxs, d_ups, d_lows, d_bits = rowquant_mix_bwd_batch(gs, xs, bits, symmetric, ups, lows, hard, threshold, want_dx=need_dx)
        out = [None, None, None, None]
        for i in range(n):
            out += [dxs[i] if need_dx else None, d_bits[i].reshape(nshapes[i])]
            if lwc:
                out += [d_ups[i], d_lows[i]]
        return tuple(out)


def reorder_index(xmax, xmin, n_heads: int = 1):
    """int64 [C] permutation: per head, channels by ascending (xmax - xmin), ties by channel id."""
    dev = L.check_gpu(xmax, xmin)
    xmax = xmax.float().contiguous()
    xmin = xmin.float().contiguous()
    C = xmax.numel()
    out = torch.empty(C, dtype=torch.int64, device=dev)
    with L.on_device(dev):
        L.call("miq_reorder_index", L.ptr(xmax), L.ptr(xmin), L.i64(C), L.i32(int(n_heads)), L.ptr(out), L.stream_of(xmax))
    return out
