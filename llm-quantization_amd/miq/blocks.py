"""Quantized decoder blocks assembled from the drop-in wrappers.  They are the objects the
block-calibration loop drives; attribute and parameter names follow the reference's
QuantOPTDecoderLayer (models/int_opt_layer.py:251-511) so that ``omni_state_dict`` produces the same
keys as the reference's ``omni_parameters.pth`` (``self_attn.q_proj.weight_quantizer.upbound_factor``,
``qkv_smooth_scale`` ...).

The attention / MLP arithmetic itself (GEMMs, softmax, activation) is plain PyTorch-ROCm: it is not
part of the fake-quantization path, only its caller.
"""
from __future__ import annotations

from collections import OrderedDict
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as TF

from . import functional as F
from .int_linear import QuantLinear
from .int_matmul import QuantMatMul
from .let import LetPlan, LinearStep, NormStep, truncate_number
from .omni_norm import OmniLayerNorm, OmniLlamaRMSNorm
from .quantizer import batched_static_quant, norm_shared_quant, shared_static_quant

TRUNCATE_THRESHOLD = 1e-2  # truncate_number's default where the decoder layers call it (models/transformation.py:18)


def quant_args(wbits=4, abits=4, w_symmetric=False, group_size=None, lwc=True, a_dynamic_method="per_cluster",
               a_dynamic=False, metric="ema_minmax", weight_exp_quant=False, weight_mix_quant=False,
               weight_mix2_quant=False, let=True):
    """Quantizer keyword dictionaries as the reference's CLI builds them (main.py:332-380)."""
    a = SimpleNamespace()
    a.weight_quant_params = {
        "n_bits": wbits if not weight_mix_quant else [], "per_channel_axes": [0], "symmetric": w_symmetric,
        "dynamic_method": "per_channel", "group_size": group_size, "lwc": lwc, "dynamic": True, "metric": "minmax",
        "target_bit": wbits,
    }
    act = {"n_bits": abits, "per_channel_axes": [], "symmetric": False, "dynamic_method": a_dynamic_method,
           "dynamic": a_dynamic, "metric": metric}
    a.act_quant_params = dict(act)
    a.q_quant_params = dict(act)
    a.k_quant_params = dict(act)
    a.v_quant_params = dict(act)
    a.p_quant_params = {"n_bits": 16, "metric": "fix0to1"}  # attention probabilities stay unquantized
    a.weight_exp_quant = weight_exp_quant
    a.weight_mix_quant = weight_mix_quant
    a.weight_mix2_quant = weight_mix2_quant
    a.let = let
    return a


def _linear(w, b):
    l = nn.Linear(w.shape[1], w.shape[0], bias=b is not None, device=w.device, dtype=w.dtype)
    with torch.no_grad():
        l.weight.copy_(w)
        if b is not None:
            l.bias.copy_(b)
    return l


class _BlockBase(nn.Module):
    """Calibration plumbing shared by the block flavours."""

    let = False

    def _quant_modules(self):
        """(name, module) of every QuantLinear / QuantMatMul, walked once: the calibration loop asks several times
        per step and the module tree does not change after construction."""
        cache = self.__dict__.get("_quant_module_cache")
        if cache is None:
            cache = [(n, m) for n, m in self.named_modules() if isinstance(m, (QuantLinear, QuantMatMul))]
            self.__dict__["_quant_module_cache"] = cache
        return cache

    def quant_linears(self):
        return [(n, m) for n, m in self._quant_modules() if isinstance(m, QuantLinear)]

    def act_quantizers(self):
        for _n, m in self._quant_modules():
            if isinstance(m, QuantLinear) and not m.disable_input_quant:
                yield m.act_quantizer
            if isinstance(m, QuantMatMul):
                yield m.x1_quantizer
                yield m.x2_quantizer

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self.use_weight_quant = weight_quant
        self.use_act_quant = act_quant
        for _n, m in self._quant_modules():
            m.set_quant_state(weight_quant, act_quant)

    def clear_temp_variable(self):
        for _n, m in self.quant_linears():
            del m.temp_weight
            del m.temp_bias

    def _named(self, *needles):
        return iter([p for n, p in self.named_parameters() if any(n.find(s) > -1 for s in needles)])

    def let_parameters(self, use_shift=True):
        return self._named("smooth" if use_shift else "smooth_scale")

    def lwc_parameters(self):
        return self._named("bound_factor")

    def nbits_parameters(self):
        return self._named("n_bits")

    def omni_parameters(self, use_shift=True):
        return self._named("bound_factor", "smooth" if use_shift else "smooth_scale")

    def omni_state_dict(self, destination=None, prefix="", keep_vars=False):
        if destination is None:
            destination = OrderedDict()
        for name, param in self.named_parameters():
            if name.find("smooth") > -1 or name.find("bound_factor") > -1:
                destination[prefix + name] = param if keep_vars else param.detach()
        return destination

    def register_scales_and_zeros(self):
        for _n, m in self.quant_linears():
            m.weight_quantizer.register_scales_and_zeros()

    def get_size_loss(self):
        mods = [m for _n, m in self.quant_linears()]
        dev = mods[0].weight.device
        model_sizes = torch.zeros((len(mods), 1), device=dev)
        target_sizes = torch.zeros((len(mods), 1), device=dev)
        for i, m in enumerate(mods):
            model_sizes[i], target_sizes[i] = m.weight_quantizer.model_size_loss()
        return model_sizes, target_sizes

    def set_hard_quant(self):
        for _n, m in self.quant_linears():
            m.weight_quantizer.set_hard()

    def round_nbits(self):
        for _n, m in self.quant_linears():
            m.weight_quantizer.round_nbits()

    def _quantize_temp_weights(self):
        """temp_weight <- weight_quantizer(temp_weight) for every linear.  Linears that use the plain uniform quantizer
        (the exponent quantizer v2, the learnable and the fixed mixed-bit quantizers) with the same settings and row width
        share one kernel launch."""
        from . import functional as F
        from .quantizer import UniformAffineQuantizer, lwc_clip_factors
        from .quantizer_exp_v2 import ExpAffineQuantizer as Exp2Quantizer
        from .quantizer_mix import MixAffineQuantizer
        from .quantizer_mix_v2 import Mix2AffineQuantizer
        groups, exp_groups, mix_groups, mix2_groups, rest = {}, {}, {}, {}, []
        for _n, m in self.quant_linears():
            src = m.temp_weight if hasattr(m, "temp_weight") else m.weight
            q = m.weight_quantizer
            if self._half_weight_path(q, src):
                m.temp_weight, _ = q.forward_let(src)
                continue
            if not self.batch_weights:
                rest.append((m, src))
                continue
            if (type(q) is Exp2Quantizer and getattr(q, "enable", True) and 2 <= q.n_bits <= 4 and not q.group_size and src.is_cuda
                    and src.dim() == 2 and src.dtype in (torch.float32, torch.float16)
                    and (not q.lwc or q.upbound_factor.dtype == src.dtype)):
                exp_groups.setdefault((src.shape[1], src.dtype, q.n_bits, bool(q.lwc)), []).append((m, src))
                continue
            if (type(q) is MixAffineQuantizer and not q.group_size and src.is_cuda and src.dim() == 2
                    and src.dtype == torch.float32 and q.n_bits.dtype == torch.float32):
                key = (src.shape[1], bool(q.symmetric), bool(q.hard_assign), float(q.threshold), bool(q.lwc))
                mix_groups.setdefault(key, []).append((m, src))
                continue
            if (type(q) is Mix2AffineQuantizer and not q.group_size and src.is_cuda and src.dim() == 2
                    and src.dtype == torch.float32 and (not q.lwc or q.upbound_factor.dtype == src.dtype)):
                mix2_groups.setdefault((src.shape[1], bool(q.symmetric), bool(q.lwc)), []).append((m, src))
                continue
            if (type(q) is UniformAffineQuantizer and q.enable and q.n_bits < 16 and q.metric != "fix0to1" and q.dynamic
                    and q.dynamic_method in ("per_token", "per_channel") and not q.group_size and src.is_cuda and src.dim() == 2
                    and src.dtype in (torch.float32, torch.float16) and (not q.lwc or q.upbound_factor.dtype == src.dtype)):
                key = (src.shape[1], src.dtype, q.n_bits, bool(q.symmetric), bool(q.lwc))
                groups.setdefault(key, []).append((m, src))
            else:
                rest.append((m, src))
        for (_i, _dt, n_bits, sym, lwc), members in groups.items():
            if len(members) == 1:
                rest.append(members[0])
                continue
            flat = []
            for m, src in members:
                flat.append(src)
                if lwc:
                    flat += list(lwc_clip_factors(m.weight_quantizer, src))
            res = F.UniformRowFQBatch.apply(n_bits, sym, lwc, *flat)
            for j, (m, _src) in enumerate(members):
                m.temp_weight = res[3 * j]
                m.weight_quantizer.scale = res[3 * j + 1]
                m.weight_quantizer.round_zero_point = res[3 * j + 2]
        for (_i, _dt, n_bits, lwc), members in exp_groups.items():
            if len(members) == 1:
                rest.append(members[0])
                continue
            flat = []
            for m, src in members:
                flat.append(src)
                if lwc:
                    flat += list(lwc_clip_factors(m.weight_quantizer, src))
            res = F.Exp2RowFQBatch.apply(n_bits, lwc, *flat)
            for j, (m, _src) in enumerate(members):
                m.temp_weight = res[2 * j]
                m.weight_quantizer.scale = res[2 * j + 1]
                m.weight_quantizer.zero_point = None
        for (_i, sym, hard, threshold, lwc), members in mix_groups.items():
            if len(members) == 1:
                rest.append(members[0])
                continue
            flat = []
            for m, src in members:
                q = m.weight_quantizer
                q.shape = src.shape
                if isinstance(q.n_bits, torch.nn.Parameter):
                    q.n_bits.data = q.n_bits.data.clamp(2, 8)  # as MixAffineQuantizer.forward does before use
                flat += [src, q.n_bits]
                if lwc:
                    flat += list(lwc_clip_factors(q, src))
            res = F.MixRowFQBatch.apply(sym, hard, threshold, lwc, *flat)
            for j, (m, _src) in enumerate(members):
                m.temp_weight = res[j]
        for (_i, sym, lwc), members in mix2_groups.items():
            if len(members) == 1:
                rest.append(members[0])
                continue
            flat, bits = [], []
            for m, src in members:
                q = m.weight_quantizer
                q.shape = src.shape
                bits.append(q._row_bits(src))
                flat.append(src)
                if lwc:
                    flat += list(lwc_clip_factors(q, src))
            res = F.Mix2RowFQBatch.apply(sym, lwc, bits, *flat)
            for j, (m, _src) in enumerate(members):
                m.temp_weight = res[j]
        for m, src in rest:
            m.temp_weight = m.weight_quantizer(src)
        for _n, m in self.quant_linears():
            if not hasattr(m, "temp_bias"):
                m.temp_bias = m.bias
            m.use_temporary_parameter = True

    batch_weights = True  # weight matrices of one width and quantizer setting share one launch each way

    half_weights = True  # under fp16 autocast, quantize plain fp32 weights straight into the fp16 tensor the linear uses

    def _half_weight_path(self, q, w) -> bool:
        """Under fp16 autocast the linear casts its fp32 quantized weight to fp16 (and autograd casts the fp16 weight
        gradient back): 6 extra bytes per element each way.  The LET-fused kernel without factors is the plain
        row-wise quantizer with an fp16 store and an fp16 gradient load, so weights that are not themselves functions
        of trained tensors go through it."""
        from .quantizer import UniformAffineQuantizer
        return (self.half_weights and self.fused_let and torch.is_autocast_enabled()
                and torch.get_autocast_dtype("cuda") == torch.float16 and type(q) is UniformAffineQuantizer
                and not w.requires_grad and w.dtype == torch.float32 and q.let_fusable(w))

    fused_let = True  # use the LET-fused weight quantizer kernels when the configuration allows

    def _let_fusable(self):
        from .quantizer import UniformAffineQuantizer
        return self.fused_let and all(type(m.weight_quantizer) is UniformAffineQuantizer and m.weight_quantizer.let_fusable(m.weight)
                                      for _n, m in self.quant_linears())

    def _let_linear(self, st: LinearStep):
        """temp_weight / temp_bias of one linear of the LET plan with the smoothed weight formed inside the quantizer
        kernel: quantize((W * col_scale) {/, *} row_scale), and W @ col_shift as a by-product for the bias."""
        m = st.linear
        plain = st.col_scale is None and st.row_scale is None and st.col_shift is None
        if plain and not self._half_weight_path(m.weight_quantizer, m.weight):
            wq, wshift = m.weight_quantizer(m.weight), None  # nothing to fuse: the plain row kernels are leaner
        else:
            wq, wshift = m.weight_quantizer.forward_let(m.weight, st.col_scale, st.row_scale, st.row_div, st.col_shift)
        m.temp_weight = wq
        m.temp_bias = st.bias(wshift)
        m.use_temporary_parameter = True

    def _smooth_and_quant_temporary(self):
        """The training-step form of the block's LET plan followed by the weight quantizers (reference
        smooth_and_quant_temporary, models/int_opt_layer.py:404-444, models/int_llama_layer.py:279-307)."""
        # the clipping factors of all weight quantizers are activated in one pass for this step
        from .quantizer import clear_lwc_activations, set_lwc_activations
        touched = set_lwc_activations([m.weight_quantizer for _n, m in self.quant_linears()])
        try:
            self._smooth_and_quant_temporary_step()
        finally:
            clear_lwc_activations(touched)

    def _smooth_and_quant_temporary_step(self):
        if self.let:
            self._truncate_scales()
            plan = self.let_plan()
            if self._let_fusable():
                plan.set_temporary_norms()
                for st in plan.linears:
                    self._let_linear(st)
                return
            plan.materialize()
        else:
            for _n, m in self.quant_linears():
                m.temp_weight = m.weight
        self._quantize_temp_weights()

    @torch.no_grad()
    def _smooth_and_quant_inplace(self):
        """The plan committed into the stored weights, then the weights quantized for good (reference
        smooth_and_quant_inplace)."""
        if self.let:
            self._truncate_scales()
            self.let_plan().commit()
        for _n, m in self.quant_linears():
            m.weight = m.weight_quantizer(m.weight)
            m.use_temporary_parameter = False

    def _truncate_scales(self):
        # the smoothing scales are parameters of the block itself (init_let registers them): the list is rebuilt only
        # when that dict changes, not by a walk over the module tree on every step
        key = tuple((n, id(q)) for n, q in self._parameters.items())  # init_let may re-register a name with a new tensor
        cache = self.__dict__.get("_smooth_scale_cache")
        if cache is None or cache[0] != key:
            cache = (key, [p for n, p in self.named_parameters() if "smooth_scale" in n])
            self.__dict__["_smooth_scale_cache"] = cache
        ps = cache[1]
        with torch.no_grad():
            # in place (the reference rebinds .data): the parameters keep their storage
            if ps and all(p.is_cuda and p.is_contiguous() and p.dtype == ps[0].dtype for p in ps) \
                    and ps[0].dtype in (torch.float32, torch.float16):
                F.truncate_inplace_([p.data for p in ps], TRUNCATE_THRESHOLD)  # one launch for all scales
            else:
                for p in ps:
                    p.copy_(truncate_number(p))


SHARE_ACT_QUANT = True  # module-wide switch (A/B timing): one activation-quantizer evaluation for layers reading one tensor


def _linears_on_one_input(linears, x, pre_norm=None):
    """[lin(x) for lin in linears] (x = pre_norm(x) first when a norm is given) with the input quantized once when the
    layers' activation quantizers cannot differ, and the quantizer's statistics gathered by the norm kernel when it
    calibrates on fresh statistics."""
    if SHARE_ACT_QUANT and all(m.use_act_quant and not m.disable_input_quant for m in linears):
        qs = [m.act_quantizer for m in linears]
        if pre_norm is not None:
            xq = norm_shared_quant(pre_norm, qs, x)
            if xq is not None:
                return [m(xq, input_quantized=True) for m in linears]
            x, pre_norm = pre_norm(x), None
        xq = shared_static_quant(qs, x)
        if xq is not None:
            return [m(xq, input_quantized=True) for m in linears]
    if pre_norm is not None:
        x = pre_norm(x)
    return [m(x) for m in linears]


def _quant_operands(matmuls_and_slots, xs):
    """Operand quantizers of the attention matmuls on tensors of one shape (q, k, v), batched into one launch pair when
    they calibrate on fresh statistics.  matmuls_and_slots: (QuantMatMul, 1 | 2) per tensor."""
    qs = [(m.x1_quantizer if slot == 1 else m.x2_quantizer) for m, slot in matmuls_and_slots]
    if SHARE_ACT_QUANT and all(m.use_act_quant for m, _ in matmuls_and_slots):
        return batched_static_quant(qs, xs)
    return [(m.quant_x1(x) if slot == 1 else m.quant_x2(x)) for (m, slot), x in zip(matmuls_and_slots, xs)]


class _OPTAttention(nn.Module):
    def __init__(self, q, k, v, o, num_heads, args, nbits=None):
        super().__init__()
        self.embed_dim = q.weight.shape[1]
        self.num_heads = num_heads
        self.head_dim = self.embed_dim // num_heads
        assert self.head_dim * num_heads == self.embed_dim
        self.scaling = self.head_dim ** -0.5
        kw = dict(weight_exp_quant=args.weight_exp_quant, weight_mix_quant=args.weight_mix_quant,
                  weight_mix2_quant=args.weight_mix2_quant)
        nb = nbits or {}
        self.k_proj = QuantLinear(k, args.weight_quant_params, args.act_quant_params, False, weight_nbits=nb.get("k", []), **kw)
        self.v_proj = QuantLinear(v, args.weight_quant_params, args.act_quant_params, False, weight_nbits=nb.get("v", []), **kw)
        self.q_proj = QuantLinear(q, args.weight_quant_params, args.act_quant_params, False, weight_nbits=nb.get("q", []), **kw)
        self.out_proj = QuantLinear(o, args.weight_quant_params, args.act_quant_params, False, weight_nbits=nb.get("o", []), **kw)
        self.qkt_matmul = QuantMatMul(args.q_quant_params, args.k_quant_params, matmul_func=torch.bmm)
        self.pv_matmul = QuantMatMul(args.p_quant_params, args.v_quant_params, matmul_func=torch.bmm)
        self.fused_softmax = True

    def _heads(self, t, bsz):
        return t.view(bsz, -1, self.num_heads, self.head_dim).transpose(1, 2).contiguous()

    def forward(self, hidden_states, attention_mask=None, pre_norm=None):
        bsz, tgt_len, _ = hidden_states.size()
        q, k, v = _linears_on_one_input([self.q_proj, self.k_proj, self.v_proj], hidden_states, pre_norm)
        q, k, v = _quant_operands([(self.qkt_matmul, 1), (self.qkt_matmul, 2), (self.pv_matmul, 2)], [q * self.scaling, k, v])
        k, v = self._heads(k, bsz), self._heads(v, bsz)
        shape = (bsz * self.num_heads, -1, self.head_dim)
        q = self._heads(q, bsz).view(*shape)
        k = k.view(*shape)
        v = v.view(*shape)
        src_len = k.size(1)
        w = self.qkt_matmul(q, k.transpose(1, 2))
        w4 = w.view(bsz, self.num_heads, tgt_len, src_len)
        # the fused pass returns the scores' dtype: usable when the mask does not promote them (fp16 scores + fp32
        # mask give fp32 probabilities in this block, unlike the Llama block which casts back)
        same_dtype = attention_mask is None or attention_mask.dtype == w.dtype
        if self.fused_softmax and same_dtype and F.masked_softmax_supported(w4, attention_mask):
            w = F.MaskedSoftmax.apply(w4, attention_mask, 1.0).view(bsz * self.num_heads, tgt_len, src_len)
        else:
            if attention_mask is not None:
                w = w4 + attention_mask
                w = w.clamp(min=torch.finfo(w.dtype).min)  # torch.max against the constant, without a host-to-device copy
                w = w.view(bsz * self.num_heads, tgt_len, src_len)
            if w.dtype == torch.float16:
                w = TF.softmax(w, dim=-1, dtype=torch.float32).to(torch.float16)
            else:
                w = TF.softmax(w, dim=-1)
        w = self.pv_matmul.quant_x1(w)
        o = self.pv_matmul(w, v)
        o = o.view(bsz, self.num_heads, tgt_len, self.head_dim).transpose(1, 2).reshape(bsz, tgt_len, self.embed_dim)
        return self.out_proj(o)


class QuantOPTBlock(_BlockBase):
    """Pre-LN OPT decoder block (layer norm -> attention -> residual; layer norm -> fc1 -> ReLU ->
    fc2 -> residual), the layout of the reference's QuantOPTDecoderLayer.

    ``weights``: dict with 2-D tensors q,k,v,o,fc1,fc2, biases q_b..fc2_b, and ln1_w, ln1_b, ln2_w,
    ln2_b.  ``nbits``: optional dict of per-row bit tensors for the fixed mixed-bit quantizer.
    """

    def __init__(self, weights: dict, num_heads: int, args, nbits=None, eps=1e-5):
        super().__init__()
        w = weights
        H = w["q"].shape[1]
        self.embed_dim = H
        self.self_attn = _OPTAttention(_linear(w["q"], w.get("q_b")), _linear(w["k"], w.get("k_b")),
                                       _linear(w["v"], w.get("v_b")), _linear(w["o"], w.get("o_b")), num_heads, args, nbits)
        self.do_layer_norm_before = True
        ln1 = nn.LayerNorm(H, eps=eps, device=w["q"].device, dtype=w["q"].dtype)
        ln2 = nn.LayerNorm(H, eps=eps, device=w["q"].device, dtype=w["q"].dtype)
        with torch.no_grad():
            ln1.weight.copy_(w["ln1_w"]); ln1.bias.copy_(w["ln1_b"])
            ln2.weight.copy_(w["ln2_w"]); ln2.bias.copy_(w["ln2_b"])
        self.self_attn_layer_norm = OmniLayerNorm(ln1)
        kw = dict(weight_exp_quant=args.weight_exp_quant, weight_mix_quant=args.weight_mix_quant,
                  weight_mix2_quant=args.weight_mix2_quant)
        nb = nbits or {}
        self.fc1 = QuantLinear(_linear(w["fc1"], w.get("fc1_b")), args.weight_quant_params, args.act_quant_params,
                               weight_nbits=nb.get("fc1", []), **kw)
        self.fc2 = QuantLinear(_linear(w["fc2"], w.get("fc2_b")), args.weight_quant_params, args.act_quant_params,
                               weight_nbits=nb.get("fc2", []), **kw)
        self.final_layer_norm = OmniLayerNorm(ln2)
        self.type = w["fc1"].dtype
        self.let = bool(getattr(args, "let", False))
        self.num_heads = num_heads

    def forward(self, hidden_states, attention_mask=None, **_kw):
        residual = hidden_states
        h = self.self_attn(hidden_states, attention_mask=attention_mask, pre_norm=self.self_attn_layer_norm)
        h = residual + h
        shape = h.shape
        h = h.reshape(-1, h.size(-1))
        residual = h
        (h,) = _linears_on_one_input([self.fc1], h, self.final_layer_norm)
        h = self.fc2(TF.relu(h))
        return ((residual + h).view(shape),)

    def let_plan(self) -> LetPlan:
        """Layer norms hand their scale / shift to q, k, v and to fc1; v_proj -> out_proj; q_proj <-> k_proj
        (models/int_opt_layer.py:404-427)."""
        a = self.self_attn
        s1, b1, so, bo, sq = (self.qkv_smooth_scale, self.qkv_smooth_shift, self.out_smooth_scale, self.out_smooth_shift,
                              self.qkt_smooth_scale)
        sf, bf = self.fc1_smooth_scale, self.fc1_smooth_shift
        return LetPlan(
            norms=[NormStep(self.self_attn_layer_norm, s1, b1), NormStep(self.final_layer_norm, sf, bf)],
            linears=[LinearStep(a.q_proj, s1, b1, sq, True), LinearStep(a.k_proj, s1, b1, sq, False),
                     LinearStep(a.v_proj, s1, b1, so, True, bo), LinearStep(a.out_proj, so, bo),
                     LinearStep(self.fc1, sf, bf), LinearStep(self.fc2)])

    def smooth_and_quant_temporary(self):
        self._smooth_and_quant_temporary()

    def smooth_and_quant_inplace(self):
        self._smooth_and_quant_inplace()

    # names of the linears that receive a LET scale/shift pair, and the reorder index each pair follows
    LET_PAIRS = {"q_proj": ("qkv", "R1"), "out_proj": ("out", "R3"), "fc1": ("fc1", "R4")}


# ------------------------------------------------------------------------------------------------
# Llama
# ------------------------------------------------------------------------------------------------
def _rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


class _Rotary(nn.Module):
    """Rotary position embedding in the Hugging Face Llama layout (the two halves of a head are the
    real / imaginary parts).  cos/sin tables are built once per (length, device, dtype)."""

    fused = True  # one kernel per tensor and direction instead of the elementwise chain (class-wide switch)

    def __init__(self, head_dim: int, theta: float = 10000.0):
        super().__init__()
        self.head_dim, self.theta = head_dim, theta
        self._tab = None

    def tables(self, position_ids, device, dtype):
        key = (position_ids.data_ptr(), position_ids._version, tuple(position_ids.shape), str(device), dtype)
        if self._tab is None or self._tab[0] != key:
            inv = 1.0 / (self.theta ** (torch.arange(0, self.head_dim, 2, dtype=torch.float32, device=device) / self.head_dim))
            freqs = position_ids.to(device=device, dtype=torch.float32)[..., None] * inv  # [b, t, hd/2]
            emb = torch.cat((freqs, freqs), dim=-1)
            self._tab = (key, emb.cos().to(dtype).unsqueeze(1), emb.sin().to(dtype).unsqueeze(1))  # [b, 1, t, hd]
        return self._tab[1], self._tab[2]

    def forward(self, q, k, position_ids):
        cos, sin = self.tables(position_ids, q.device, q.dtype)
        if self.fused and k.dtype == q.dtype and F.rotary_supported(q, cos, sin) and F.rotary_supported(k, cos, sin):
            return F.RotaryEmb.apply(q, cos, sin), F.RotaryEmb.apply(k, cos, sin)
        return q * cos + _rotate_half(q) * sin, k * cos + _rotate_half(k) * sin


class _LlamaAttention(nn.Module):
    """reference models/int_llama_layer.py:48-179 (no KV cache: calibration always sees whole sequences)."""

    def __init__(self, q, k, v, o, num_heads, num_kv_heads, args, rope_theta=10000.0, nbits=None):
        super().__init__()
        self.hidden_size = q.weight.shape[1]
        self.num_heads = num_heads
        self.head_dim = self.hidden_size // num_heads
        self.num_key_value_heads = num_kv_heads
        self.num_key_value_groups = num_heads // num_kv_heads
        if self.head_dim * num_heads != self.hidden_size:
            raise ValueError(f"hidden_size {self.hidden_size} must be divisible by num_heads {num_heads}")
        self.rotary_emb = _Rotary(self.head_dim, rope_theta)
        kw = dict(weight_exp_quant=args.weight_exp_quant, weight_mix_quant=args.weight_mix_quant,
                  weight_mix2_quant=args.weight_mix2_quant)
        nb = nbits or {}
        self.k_proj = QuantLinear(k, args.weight_quant_params, args.act_quant_params, weight_nbits=nb.get("k", []), **kw)
        self.v_proj = QuantLinear(v, args.weight_quant_params, args.act_quant_params, weight_nbits=nb.get("v", []), **kw)
        self.q_proj = QuantLinear(q, args.weight_quant_params, args.act_quant_params, weight_nbits=nb.get("q", []), **kw)
        self.o_proj = QuantLinear(o, args.weight_quant_params, args.act_quant_params, weight_nbits=nb.get("o", []), **kw)
        self.qkt_matmul = QuantMatMul(args.q_quant_params, args.k_quant_params, matmul_func=torch.matmul)
        self.pv_matmul = QuantMatMul(args.p_quant_params, args.v_quant_params, matmul_func=torch.matmul)
        self.fused_softmax = True

    def forward(self, hidden_states, attention_mask=None, position_ids=None, pre_norm=None):
        bsz, q_len, _ = hidden_states.size()
        q, k, v = _linears_on_one_input([self.q_proj, self.k_proj, self.v_proj], hidden_states, pre_norm)
        q = q.view(bsz, q_len, self.num_heads, self.head_dim).transpose(1, 2)
        k = k.view(bsz, q_len, self.num_key_value_heads, self.head_dim).transpose(1, 2)
        v = v.view(bsz, q_len, self.num_key_value_heads, self.head_dim).transpose(1, 2)
        if position_ids is None:
            position_ids = torch.arange(q_len, device=hidden_states.device).unsqueeze(0)
        q, k = self.rotary_emb(q, k, position_ids)
        if self.num_key_value_groups > 1:
            k = k[:, :, None].expand(bsz, self.num_key_value_heads, self.num_key_value_groups, q_len, self.head_dim)
            k = k.reshape(bsz, self.num_heads, q_len, self.head_dim)
            v = v[:, :, None].expand(bsz, self.num_key_value_heads, self.num_key_value_groups, q_len, self.head_dim)
            v = v.reshape(bsz, self.num_heads, q_len, self.head_dim)
        q, k, v = _quant_operands([(self.qkt_matmul, 1), (self.qkt_matmul, 2), (self.pv_matmul, 2)], [q, k, v])
        w = self.qkt_matmul(q, k.transpose(2, 3))
        if self.fused_softmax and F.masked_softmax_supported(w, attention_mask):
            # scale, mask, floor, fp32 softmax and the cast back in one pass over the score matrix
            w = F.MaskedSoftmax.apply(w, attention_mask, self.head_dim ** 0.5)
        else:
            w = w / (self.head_dim ** 0.5)
            if attention_mask is not None:
                w = w + attention_mask
                w = w.clamp(min=torch.finfo(w.dtype).min)  # torch.max against the constant, without a host-to-device copy
            w = TF.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
        w = self.pv_matmul.quant_x1(w)
        o = self.pv_matmul(w, v)
        o = o.transpose(1, 2).reshape(bsz, q_len, self.hidden_size)
        return self.o_proj(o)


class _LlamaMLP(nn.Module):
    def __init__(self, gate, up, down, args, nbits=None):
        super().__init__()
        kw = dict(weight_exp_quant=args.weight_exp_quant, weight_mix_quant=args.weight_mix_quant,
                  weight_mix2_quant=args.weight_mix2_quant)
        nb = nbits or {}
        self.gate_proj = QuantLinear(gate, args.weight_quant_params, args.act_quant_params, weight_nbits=nb.get("gate", []), **kw)
        self.down_proj = QuantLinear(down, args.weight_quant_params, args.act_quant_params, weight_nbits=nb.get("down", []), **kw)
        self.up_proj = QuantLinear(up, args.weight_quant_params, args.act_quant_params, weight_nbits=nb.get("up", []), **kw)

    def forward(self, x, pre_norm=None):
        gate, up = _linears_on_one_input([self.gate_proj, self.up_proj], x, pre_norm)
        return self.down_proj(TF.silu(gate) * up)


class QuantLlamaBlock(_BlockBase):
    """Llama decoder block (RMSNorm -> attention with rotary embedding and grouped KV heads ->
    residual; RMSNorm -> SwiGLU MLP -> residual), the layout of the reference's
    QuantLlamaDecoderLayer (models/int_llama_layer.py:191-368).

    ``weights``: dict with 2-D tensors q, k, v, o, gate, up, down (optional biases ``<name>_b``) and
    the norm weights ln1_w, ln2_w.
    """

    LET_PAIRS = {"q_proj": ("qkv", "R1"), "o_proj": ("out", "R3"), "up_proj": ("fc1", "R4")}

    def __init__(self, weights: dict, num_heads: int, args, num_kv_heads=None, nbits=None, eps=1e-6, rope_theta=10000.0):
        super().__init__()
        w = weights
        self.hidden_size = w["q"].shape[1]
        self.self_attn = _LlamaAttention(_linear(w["q"], w.get("q_b")), _linear(w["k"], w.get("k_b")),
                                         _linear(w["v"], w.get("v_b")), _linear(w["o"], w.get("o_b")), num_heads,
                                         num_kv_heads or num_heads, args, rope_theta, nbits)
        self.mlp = _LlamaMLP(_linear(w["gate"], w.get("gate_b")), _linear(w["up"], w.get("up_b")),
                             _linear(w["down"], w.get("down_b")), args, nbits)
        n1, n2 = SimpleNamespace(weight=w["ln1_w"].detach().clone()), SimpleNamespace(weight=w["ln2_w"].detach().clone())
        self.input_layernorm = OmniLlamaRMSNorm(n1, eps=eps)
        self.post_attention_layernorm = OmniLlamaRMSNorm(n2, eps=eps)
        self.let = bool(getattr(args, "let", False))
        self.num_heads = num_heads

    def forward(self, hidden_states, attention_mask=None, position_ids=None, **_kw):
        residual = hidden_states
        h = self.self_attn(hidden_states, attention_mask=attention_mask, position_ids=position_ids, pre_norm=self.input_layernorm)
        h = residual + h
        residual = h
        h = self.mlp(h, pre_norm=self.post_attention_layernorm)
        return (residual + h,)

    def let_plan(self) -> LetPlan:
        """RMSNorms hand their scale / shift to q, k, v and to up / gate; v_proj -> o_proj; q_proj <-> k_proj
        (models/int_llama_layer.py:279-296)."""
        a, m = self.self_attn, self.mlp
        s1, b1, so, bo, sq = (self.qkv_smooth_scale, self.qkv_smooth_shift, self.out_smooth_scale, self.out_smooth_shift,
                              self.qkt_smooth_scale)
        sf, bf = self.fc1_smooth_scale, self.fc1_smooth_shift
        return LetPlan(
            norms=[NormStep(self.input_layernorm, s1, b1), NormStep(self.post_attention_layernorm, sf, bf)],
            linears=[LinearStep(a.q_proj, s1, b1, sq, True), LinearStep(a.k_proj, s1, b1, sq, False),
                     LinearStep(a.v_proj, s1, b1, so, True, bo), LinearStep(a.o_proj, so, bo),
                     LinearStep(m.up_proj, sf, bf), LinearStep(m.gate_proj, sf, bf), LinearStep(m.down_proj)])

    def smooth_and_quant_temporary(self):
        self._smooth_and_quant_temporary()

    def smooth_and_quant_inplace(self):
        self._smooth_and_quant_inplace()
