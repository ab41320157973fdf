"""UniformAffineQuantizer: same constructor, attributes and methods as the reference class
(quantize/quantizer.py:21-283), with the arithmetic done by libmiq's fused HIP kernels.

Two regimes, as in the reference:
  * ``dynamic=True`` (weights; activations with --a_dynamic): row-wise min/max over the last axis,
    optional learnable weight clipping, one fused kernel (forward) + one (backward);
  * ``dynamic=False`` (static activations): calibration mode accumulates per-tensor or per-cluster
    statistics (running min/max or EMA) AND quantizes; eval mode quantizes with the frozen buffers.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import functional as F
from ._lib import MiqError

CLIPMIN = 1e-5


def round_ste(x: torch.Tensor):
    """Round with a straight-through gradient (kept for API compatibility)."""
    return (x.round() - x).detach() + x


def lwc_clip_factors(q, x):
    """(sigmoid(upbound_factor), sigmoid(lowbound_factor)) of a row-wise quantizer, or (None, None) without learnable
    clipping: the tensors the reference forms at quantize/quantizer.py:259-261, evaluated by the same ATen routine, which
    is what the kernels take.  A block that quantizes all of its weights in one step evaluates the factors of all its
    quantizers in one pass (set_lwc_activations) and leaves them in ``_lwc_act`` for the duration of that step."""
    if not q.lwc:
        return None, None
    up, low = q.upbound_factor, q.lowbound_factor
    if up.dtype != x.dtype:
        raise MiqError(f"LWC factors are {up.dtype} but the tensor is {x.dtype}: call .half()/.float() on the module")
    act = q.__dict__.get("_lwc_act")
    if act is not None:
        return act
    su, sl = F.lwc_activate([up, low])
    return su, sl


def set_lwc_activations(quantizers):
    """One LwcSigmoid pass over the clipping factors of every quantizer in ``quantizers`` that has them (grouped by dtype
    and device); each keeps its pair in ``_lwc_act`` until clear_lwc_activations.  Returns the quantizers touched."""
    groups = {}
    for q in quantizers:
        if getattr(q, "lwc", False) and q.upbound_factor.is_cuda:
            groups.setdefault((q.upbound_factor.dtype, q.upbound_factor.device), []).append(q)
    touched = []
    for qs in groups.values():
        acts = F.lwc_activate([f for q in qs for f in (q.upbound_factor, q.lowbound_factor)])
        for i, q in enumerate(qs):
            q.__dict__["_lwc_act"] = (acts[2 * i], acts[2 * i + 1])
            touched.append(q)
    return touched


def clear_lwc_activations(quantizers):
    for q in quantizers:
        q.__dict__.pop("_lwc_act", None)


class UniformAffineQuantizer(nn.Module):
    def __init__(
        self,
        n_bits: int = 8,
        symmetric: bool = False,
        per_channel_axes=[],
        metric="minmax",
        dynamic=False,
        dynamic_method="per_cluster",
        group_size=None,
        shape=None,
        lwc=False,
        target_bit=0,
    ):
        super().__init__()
        self.symmetric = symmetric
        assert 2 <= n_bits <= 16, "bitwidth not supported"
        self.n_bits = n_bits
        self.qmin = 0
        self.qmax = 2 ** (n_bits) - 1
        self.per_channel_axes = per_channel_axes
        self.metric = metric
        self.cluster_counts = None
        self.cluster_dim = None

        self.scale = None
        self.zero_point = None
        self.round_zero_point = None

        self.cached_xmin = None
        self.cached_xmax = None
        self.dynamic = dynamic
        self.dynamic_method = dynamic_method
        self.deficiency = 0
        self.lwc = lwc
        self.mode = "calibration"

        init_value = 4.0  # sigmoid(4) = 0.982: start almost un-clipped
        if lwc:
            if group_size:
                dim1 = int(shape[0] * math.ceil(shape[1] / group_size))
                self.deficiency = shape[-1] % group_size
                if self.deficiency > 0:
                    self.deficiency = group_size - self.deficiency
                    assert self.symmetric  # zero padding is only meaningful for symmetric grids
            else:
                dim1 = shape[0]
            self.upbound_factor = nn.Parameter(torch.ones((dim1, 1)) * init_value)
            self.lowbound_factor = nn.Parameter(torch.ones((dim1, 1)) * init_value)
        self.sigmoid = nn.Sigmoid()

        self.enable = True
        self.group_size = group_size
        self._vec = None  # per-channel [C] copies of (scale, rzp) for the kernels
        self.dp = None    # optional miq.dist.DPContext: global-batch statistics under data parallelism

    # ------------------------------------------------------------------ small API surface
    def change_n_bits(self, n_bits):
        self.n_bits = n_bits
        self.qmin = 0
        self.qmax = 2 ** (n_bits) - 1

    def set_calibration_mode(self):
        self.mode = "calibration"

    def set_eval_mode(self):
        self.mode = "eval"

    def free(self):
        self.cached_xmax = None
        self.cached_xmin = None

    def register_scales_and_zeros(self):
        self.register_buffer("scales", self.scale)
        self.register_buffer("zeros", self.round_zero_point)
        del self.scale
        del self.round_zero_point

    # ------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor):
        if self.n_bits >= 16 or not self.enable:
            return x
        if self.metric == "fix0to1":
            # unreachable in the reference's flows (its only user is built with n_bits=16)
            return x.mul_(2 ** self.n_bits - 1).round_().div_(2 ** self.n_bits - 1)
        if self.dynamic:
            if self.dynamic_method in ("per_token", "per_channel"):
                return self._row_dynamic(x)
            raise MiqError(f"dynamic quantization with dynamic_method={self.dynamic_method!r} has no scale rule "
                           "(the reference leaves scale unset in this configuration)")
        if self.mode == "calibration":
            return self._static_calibrate_and_quant(x)
        return self._static_eval(x)

    # ------------------------------------------------------------------ dynamic, row-wise
    def _lwc_params(self, x):
        return lwc_clip_factors(self, x)

    def _row_dynamic(self, x):
        up, low = self._lwc_params(x)
        if self.group_size:
            assert x.dim() == 2, "only support linear layer now"
        out, scale, rzp = F.UniformRowFQ.apply(x, up, low, self.n_bits, bool(self.symmetric), int(self.group_size or 0),
                                               None, False)
        if not self.group_size and x.dim() != 2:
            scale = scale.reshape(*x.shape[:-1], 1)
            rzp = rzp.reshape(*x.shape[:-1], 1)
        self.scale = scale
        self.round_zero_point = rzp
        return out

    def per_token_dynamic_calibration(self, x):
        """Sets ``scale`` / ``round_zero_point`` from x (reference quantizer.py:248-273)."""
        with torch.no_grad():
            self._row_dynamic(x)

    def let_fusable(self, w) -> bool:
        """True when forward_let() may replace `forward((w * cs) / rs)` for this quantizer and weight."""
        return (self.enable and self.n_bits < 16 and self.metric != "fix0to1" and self.dynamic
                and self.dynamic_method in ("per_token", "per_channel") and F.let_supported(w, self.group_size or 0))

    def forward_let(self, w, cs=None, rs=None, rs_div=True, shift=None):
        """Quantize the smoothed weight (w * cs[None, :]) / rs[:, None] (``rs_div``) or * rs[:, None]
        without materialising it, and return it with w @ shift (None without ``shift``).  Same result
        as forward() on the materialised tensor; gradients reach cs, rs, shift and the LWC factors."""
        up, low = self._lwc_params(w)
        # under autocast the consuming linear casts the weight to fp16: let the kernel store that directly
        half_out = torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.float16
        out, wshift, scale, rzp = F.LetRowFQ.apply(w, cs, rs, bool(rs_div), shift, up, low, self.n_bits, bool(self.symmetric),
                                                   half_out)
        self.scale = scale
        self.round_zero_point = rzp
        return out, (wshift if shift is not None else None)

    # ------------------------------------------------------------------ static activations
    def _layout(self, x):
        C = x.shape[-1]
        if self.dynamic_method == "per_tensor" or self.cluster_dim is None:
            counts, kind = None, "tensor"
        else:
            if self.cluster_dim != x.dim() - 1:
                raise MiqError(f"cluster_dim={self.cluster_dim} must be the last axis of a {x.dim()}-D activation")
            counts, kind = self.cluster_counts, "cluster"
        return F.ClusterLayout.get(counts, C, x.device), kind

    def _stat_shape(self, x, kind):
        if self.dynamic_method == "per_tensor":
            return ()
        if kind == "tensor":  # per_cluster without a cluster axis: every axis reduced, keepdim
            return (1,) * x.dim()
        return (1,) * (x.dim() - 1) + (x.shape[-1],)

    def _publish(self, x, st, kind):
        shape = self._stat_shape(x, kind)
        pick = (lambda v: v[0].reshape(shape)) if kind == "tensor" else (lambda v: v.reshape(shape))
        for name, key in (("scale", "scale"), ("zero_point", "zp"), ("round_zero_point", "rzp")):
            if name in self._buffers:
                self._buffers[name] = pick(st[key])  # already a buffer: rebinding it is a dict store
                continue
            if hasattr(self, name):
                delattr(self, name)
            self.register_buffer(name, pick(st[key]))
        # plain attributes (never parameters or buffers): skip nn.Module.__setattr__'s registration checks, this runs
        # once per activation quantizer call
        d = self.__dict__
        d["cached_xmin"] = pick(st["cmin"])
        d["cached_xmax"] = pick(st["cmax"])
        d["_vec"] = (st["scale"], st["rzp"], st["cmin"], st["cmax"])

    def _cached_vectors(self, x):
        """[C] vectors of the cached statistics (or None, None)."""
        if self.cached_xmax is None:
            return None, None
        C = x.shape[-1]
        if self._vec is not None and self._vec[2].dtype == x.dtype and self._vec[2].numel() == C:
            return self._vec[2], self._vec[3]
        cm, cx = self.cached_xmin.to(x.dtype), self.cached_xmax.to(x.dtype)
        return cm.reshape(-1).expand(C).contiguous(), cx.reshape(-1).expand(C).contiguous()

    def _static_calibrate_and_quant(self, x):
        if self.symmetric:
            raise MiqError("static activation quantization is asymmetric only (the reference has no symmetric rule)")
        if self.dynamic_method not in ("per_tensor", "per_cluster"):
            raise MiqError(f"unknown dynamic_method {self.dynamic_method!r}")
        layout, kind = self._layout(x)
        cmin, cmax = self._cached_vectors(x)
        holder = {}
        out = F.ActCalibFQ.apply(x, self.n_bits, self.metric, layout, cmin, cmax, holder, self.dp)
        self._publish(x, holder, kind)
        return out

    def calibration(self, x: torch.Tensor):
        """Statistics update only (reference quantizer.py:157-168)."""
        layout, kind = self._layout(x)
        cmin, cmax = self._cached_vectors(x)
        xc = x.detach().contiguous()
        st = F.act_calibrate(xc.reshape(-1, x.shape[-1]), self.n_bits, self.metric, layout, cmin, cmax, dp=self.dp)
        self._publish(x, st, kind)

    def _eval_vectors(self, x):
        C = x.shape[-1]
        v = self._vec
        if v is not None and v[0].dtype == x.dtype and v[0].numel() == C and v[0].device == x.device:
            return v[0], v[1]
        scale = self.scale if self.scale is not None else getattr(self, "scales", None)
        rzp = self.round_zero_point if self.round_zero_point is not None else getattr(self, "zeros", None)
        if scale is None or rzp is None:
            raise MiqError("eval-mode activation quantizer has no calibrated scale")
        s = scale.to(device=x.device, dtype=x.dtype).reshape(-1).expand(C).contiguous()
        z = rzp.to(device=x.device, dtype=x.dtype).reshape(-1).expand(C).contiguous()
        self._vec = (s, z, s, z)
        return s, z

    def _static_eval(self, x):
        s, z = self._eval_vectors(x)
        return F.ActEvalFQ.apply(x, s, z, self.n_bits)

    def fake_quant(self, x, scale, round_zero_point):
        """Quantize-dequantize with given parameters (reference quantizer.py:99-121).  The layout is read from the
        SHAPE of ``scale``: one element (per tensor), last axis = channels with every other axis 1 (per channel, what
        the static calibration leaves in ``self.scale``), or one value per row / per group with a trailing axis of
        1 (what per_token_dynamic_calibration leaves).  A quantizer whose rows are zero-padded to a multiple of the group
        size (``deficiency`` > 0) pads, quantizes and strips like the reference (:100-102, :119-120).  Anything else is
        refused rather than guessed."""
        C = x.shape[-1]
        if scale.shape != round_zero_point.shape and round_zero_point.numel() != 1:
            raise MiqError(f"scale {tuple(scale.shape)} and zero point {tuple(round_zero_point.shape)} differ in shape")
        if scale.numel() == 1 or (scale.shape[-1] == C and scale.numel() == C and C != 1):
            s = scale.to(x.dtype).reshape(-1).expand(C).contiguous()
            z = round_zero_point.to(x.dtype).reshape(-1).expand(C).contiguous()
            return F.ActEvalFQ.apply(x, s, z, self.n_bits)
        pad = int(self.deficiency) if (self.group_size and self.deficiency > 0) else 0
        xp = x
        if pad:
            if x.dim() != 2:
                raise MiqError("a padded group layout is defined for 2-D weights only")
            xp = torch.cat((x, x.new_zeros((x.shape[0], pad))), dim=1)
        g = int(self.group_size or C)
        rows = xp.numel() // g if g else 0
        if scale.dim() >= 1 and scale.shape[-1] == 1 and scale.numel() == rows and xp.numel() == rows * g:
            # row-wise parameters: the per-channel kernel on the transposed slab (channels <-> rows)
            xt = xp.reshape(rows, g).t().contiguous()
            s = scale.to(x.dtype).reshape(-1).contiguous()
            z = round_zero_point.to(x.dtype).reshape(-1).expand(rows).contiguous()
            out = F.ActEvalFQ.apply(xt, s, z, self.n_bits).t().reshape(xp.shape)
            return out[:, :-pad] if pad else out
        raise MiqError(f"fake_quant: a scale of shape {tuple(scale.shape)} does not describe a per-tensor, per-channel "
                       f"([..., {C}]) or per-row ([{rows}, 1]) layout of a tensor of shape {tuple(x.shape)}")

    def _apply(self, fn, *a, **k):
        # .half()/.float()/.to(): drop the private kernel-side copies, they are rebuilt lazily
        self._vec = None
        if self.cached_xmin is not None:
            self.cached_xmin = fn(self.cached_xmin)
            self.cached_xmax = fn(self.cached_xmax)
        return super()._apply(fn, *a, **k)


# ------------------------------------------------------------------------------------------------
# several static activation quantizers evaluated together
# ------------------------------------------------------------------------------------------------
_CONFIG = ("n_bits", "symmetric", "metric", "dynamic", "dynamic_method", "mode", "enable", "cluster_dim")


def _same_config(qs):
    q0 = qs[0]
    for q in qs[1:]:
        if type(q) is not UniformAffineQuantizer or any(getattr(q, a) != getattr(q0, a) for a in _CONFIG):
            return False
        c0, c1 = q0.cluster_counts, q.cluster_counts
        if (c0 is None) != (c1 is None) or (c0 is not None and (len(c0) != len(c1) or any(int(a) != int(b) for a, b in zip(c0, c1)))):
            return False
        if q.dp is not q0.dp:
            return False
    return type(q0) is UniformAffineQuantizer and not q0.dynamic and q0.metric != "fix0to1"


def shared_static_quant(quantizers, x):
    """The result of every quantizer in ``quantizers`` applied to the SAME tensor ``x``, evaluated once when they
    cannot differ: same configuration and the same state objects (no cached statistics yet, or the ones a previous
    shared call left in all of them; in eval mode the same scale buffers).  The reference calls three quantizers on one
    layer-norm output (models/int_llama_layer.py:110-116, models/int_opt_layer.py:175-186) and gets three identical
    tensors; their backward passes add up to the backward of one (it is linear in the incoming gradient).
    Returns None when the quantizers have to be called one by one."""
    q0 = quantizers[0]
    if len(quantizers) < 2 or not _same_config(quantizers) or q0.n_bits >= 16 or not q0.enable:
        return None
    if q0.mode == "calibration":
        if any(q.cached_xmax is not q0.cached_xmax or q.cached_xmin is not q0.cached_xmin for q in quantizers[1:]):
            return None
    else:
        v0 = q0._vec
        if v0 is None or any(q._vec is None or q._vec[0] is not v0[0] or q._vec[1] is not v0[1] for q in quantizers[1:]):
            return None
    out = q0(x)
    if q0.mode == "calibration":
        _adopt_state(q0, quantizers[1:])
    return out


# Module-wide switches: statistics of a norm's output gathered by the norm kernel itself.  On by default for per-tensor
# statistics only: there the fused pair saves the statistics pass at no cost (33 us against 34.5 us for a [2048, 4096]
# fp32 slab), while with 32 clusters a row's 64 atomics and segmented exchanges cost more than the statistics kernel
# they replace (37 us against 34 us), so clustered layouts keep the separate kernels unless asked otherwise.
FUSE_NORM_QUANT = True
FUSE_NORM_QUANT_CLUSTERS = False


def _adopt_state(q0, others):
    """The quantizers of a shared evaluation hold the SAME state objects afterwards (shared_static_quant relies on it)."""
    for q in others:
        for name in ("scale", "zero_point", "round_zero_point"):
            if name not in q._buffers:
                if hasattr(q, name):
                    delattr(q, name)
                q.register_buffer(name, q0._buffers[name])
            else:
                q._buffers[name] = q0._buffers[name]
        d = q.__dict__
        d["cached_xmin"], d["cached_xmax"], d["_vec"] = q0.cached_xmin, q0.cached_xmax, q0._vec


def norm_shared_quant(norm, quantizers, x):
    """q(norm(x)) for one static quantizer, or several that cannot differ, calibrating on FRESH statistics (the training
    step): the norm kernel folds the minima / maxima of its own output into the statistics table and the quantize kernel
    follows, so the norm output is not read a second time for its statistics (quantize/omni_norm.py:29-42,60-71 followed
    by quantize/quantizer.py:124-142 in the reference).  Returns None when this form does not apply; the caller then
    evaluates norm and quantizers one after the other."""
    from .omni_norm import OmniLayerNorm, OmniLlamaRMSNorm
    q0 = quantizers[0]
    if not FUSE_NORM_QUANT or not x.is_cuda or x.numel() == 0 or x.dim() not in (2, 3):
        return None
    if type(q0) is not UniformAffineQuantizer or (len(quantizers) > 1 and not _same_config(quantizers)):
        return None
    if (q0.dynamic or q0.metric == "fix0to1" or q0.n_bits >= 16 or not q0.enable or q0.symmetric or q0.mode != "calibration"
            or q0.dynamic_method not in ("per_tensor", "per_cluster") or any(q.cached_xmax is not None for q in quantizers)
            or (q0.dp is not None and q0.dp.active and q0.dp.sync_stats)):
        return None
    if q0.dynamic_method == "per_cluster" and q0.cluster_dim is not None and q0.cluster_dim != x.dim() - 1:
        return None
    weight = norm.temp_weight if norm.use_temporary_parameter else norm.weight
    bias = norm.temp_bias if norm.use_temporary_parameter else getattr(norm, "bias", None)
    layout, kind = q0._layout(x)
    if not FUSE_NORM_QUANT_CLUSTERS and not (layout.n_clusters == 1 and layout.covered == x.shape[-1]):
        return None
    holder = {}
    if isinstance(norm, OmniLlamaRMSNorm):
        if not (norm.fused and F.rmsnorm_supported(x, weight, bias)) or x.shape[-1] % (8 if x.dtype == torch.float16 else 4):
            return None
        out = F.RMSNormActQuant.apply(x, weight.float(), None if bias is None else bias.float(), norm.variance_epsilon,
                                      q0.n_bits, layout, holder)
    elif isinstance(norm, OmniLayerNorm):
        if not (hasattr(norm, "reorder_index") and norm.fused_gather and tuple(norm.normalized_shape) == (x.shape[-1],)
                and F.gather_layernorm_supported(x, weight, bias)):
            return None
        amp = torch.is_autocast_enabled()
        if not (amp or weight is None or weight.dtype == x.dtype):
            return None
        out_dtype = torch.float32 if amp else (weight.dtype if weight is not None else x.dtype)
        if x.shape[-1] % (8 if out_dtype == torch.float16 else 4):
            return None
        i32, _inv = norm._permutation(x.device)
        out = F.GatherLayerNormActQuant.apply(x, i32, weight, bias, norm.eps, out_dtype, q0.n_bits, layout, holder)
    else:
        return None
    q0._publish(x, holder, kind)
    _adopt_state(q0, quantizers[1:])
    return out


def batched_static_quant(quantizers, xs):
    """[q(x) for q, x in zip(quantizers, xs)] for static quantizers of one configuration on tensors of one shape and
    dtype, through one pair of kernel launches when every one of them calibrates on fresh statistics (the q / k / v
    projection outputs of a training step); anything else falls back to one call per quantizer."""
    q0 = quantizers[0]
    ok = (len(quantizers) > 1 and _same_config(quantizers) and q0.n_bits < 16 and q0.enable and q0.mode == "calibration"
          and all(q.cached_xmax is None for q in quantizers) and not (q0.dp is not None and q0.dp.active and q0.dp.sync_stats)
          and all(x.shape == xs[0].shape and x.dtype == xs[0].dtype and x.is_cuda for x in xs) and xs[0].numel() > 0
          and not q0.symmetric and q0.dynamic_method in ("per_tensor", "per_cluster"))
    if not ok:
        return [q(x) for q, x in zip(quantizers, xs)]
    layout, kind = q0._layout(xs[0])
    holders = [{} for _ in quantizers]
    outs = F.ActCalibFQBatch.apply(q0.n_bits, layout, holders, *xs)
    for q, x, h in zip(quantizers, xs, holders):
        q._publish(x, h, kind)
    return list(outs)

