"""Learnable equivalent transformations (LET) as a migration plan.

A decoder block moves per-channel scales / shifts between neighbouring operators so that activations become easier
to quantize while the block computes the same function (reference models/transformation.py:24-113, driven from
models/int_opt_layer.py:404-444 and models/int_llama_layer.py:279-307):

  * a norm hands (x - shift) / scale to the linears that read it, which multiply their input columns by scale and take
    W @ shift into their bias;
  * v_proj -> out_proj: v_proj's output rows are shifted and divided, out_proj's input columns take them back;
  * q_proj <-> k_proj: q rows divided, k rows multiplied by one vector (their product q.k^T is unchanged).

Here a block states these relations ONCE as a LetPlan: per norm its (scale, shift), per linear what happens to its
input columns and to its output rows.  The same plan is then
  * materialised into temporary weights / biases that stay functions of the trained vectors (a training step),
  * committed into the stored weights (after training), or
  * handed to the LET-fused weight quantizer kernel, which forms (W * col_scale) / row_scale on the fly
(blocks._BlockBase).  Per element the arithmetic is the reference's, in its order: W * col_scale first, then the row
factor; bias = ((b + W @ col_shift) - row_shift) {/, *} row_scale.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import torch


class _Truncate(torch.autograd.Function):
    """|x| < threshold -> sign(x)*threshold, identity gradient."""

    @staticmethod
    def forward(ctx, x, threshold):
        # torch.where instead of the reference's masked assignment: same values, no host synchronisation
        return torch.where(x.abs() < threshold, x.sign() * threshold, x)

    @staticmethod
    def backward(ctx, g):
        return g.clone(), None


def truncate_number(number, threshold=1e-2):
    """Keeps smoothing scales away from zero (threshold 1e-2, as the decoder layers of the reference see it)."""
    return _Truncate.apply(number, threshold)


class _MulColumns(torch.autograd.Function):
    """w * c[None, :] for a 2-D weight and a per-column vector.  Autograd's own backward sums g * w over the rows with
    ATen's multi-block reduction; here that column sum is the library's streaming kernel (fp32 partials, fixed-order
    fold), which also keeps the step free of the one kind of ATen kernel that does not survive HIP-graph replay."""

    @staticmethod
    def forward(ctx, w, c):
        ctx.save_for_backward(w, c)
        return w * c.view(1, -1)

    @staticmethod
    def backward(ctx, g):
        from . import functional as F
        w, c = ctx.saved_tensors
        gw = g * c.view(1, -1) if ctx.needs_input_grad[0] else None
        gc = None
        if ctx.needs_input_grad[1]:
            prod = (g * w).contiguous()
            if prod.is_cuda and prod.dtype in (torch.float16, torch.float32) and prod.numel() > 0:
                gc = F.col_sum(prod).to(c.dtype).view(c.shape)
            else:
                gc = prod.sum(0).view(c.shape)
        return gw, gc


def mul_columns(w, c):
    return _MulColumns.apply(w, c) if w.dim() == 2 and c.numel() == w.shape[1] else w * c.view(1, -1)


@dataclass
class NormStep:
    """The norm emits (x - shift) / scale."""
    norm: torch.nn.Module
    scale: torch.Tensor
    shift: torch.Tensor


@dataclass
class LinearStep:
    """What the migrations do to one linear layer y = x W^T + b."""
    linear: torch.nn.Module
    col_scale: Optional[torch.Tensor] = None   # input columns multiplied by it (the producer divided by it)
    col_shift: Optional[torch.Tensor] = None   # the producer subtracted it: b += W @ col_shift
    row_scale: Optional[torch.Tensor] = None   # output rows divided (row_div) or multiplied by it
    row_div: bool = True
    row_shift: Optional[torch.Tensor] = None   # subtracted from the bias before the row factor

    def bias(self, wshift):
        """The migrated bias, given W @ col_shift (or None); None when there is nothing to add to no bias."""
        b = self.linear.bias if getattr(self.linear, "bias", None) is not None else None
        if wshift is not None:
            b = wshift if b is None else b + wshift
        if b is None:
            return None
        if self.row_shift is not None:
            b = b - self.row_shift
        if self.row_scale is not None:
            r = self.row_scale.view(-1)
            b = b / r if self.row_div else b * r
        return b

    def weight(self):
        w = self.linear.weight
        if self.col_scale is not None:
            w = mul_columns(w, self.col_scale)
        if self.row_scale is not None:
            r = self.row_scale.view(-1, 1)
            w = w / r if self.row_div else w * r
        return w

    def input_shift_product(self):
        return self.linear.weight @ self.col_shift if self.col_shift is not None else None


def _norm_params(step: NormStep):
    n = step.norm
    has_bias = getattr(n, "bias", None) is not None
    bias = (n.bias - step.shift) / step.scale if has_bias else (-1 * step.shift) / step.scale
    return n.weight / step.scale, bias


@dataclass
class LetPlan:
    norms: List[NormStep] = field(default_factory=list)
    linears: List[LinearStep] = field(default_factory=list)

    def set_temporary_norms(self):
        for st in self.norms:
            st.norm.temp_weight, st.norm.temp_bias = _norm_params(st)
            st.norm.use_temporary_parameter = True

    def materialize(self):
        """Temporary (differentiable) weights and biases of every operator in the plan; the weights are NOT yet
        quantized (the caller runs the weight quantizers over ``temp_weight``)."""
        self.set_temporary_norms()
        for st in self.linears:
            m = st.linear
            m.temp_weight = st.weight()
            m.temp_bias = st.bias(st.input_shift_product())
            m.use_temporary_parameter = True

    @torch.no_grad()
    def commit(self):
        """Write the migrated weights / biases into the operators themselves (the trained vectors become constants)."""
        for st in self.norms:
            w, b = _norm_params(st)
            _store(st.norm, "weight", w)
            _store(st.norm, "bias", b)
            st.norm.use_temporary_parameter = False
        for st in self.linears:
            m = st.linear
            b = st.bias(st.input_shift_product())  # from the stored weight, before it changes
            w = st.weight()
            _store(m, "weight", w)
            if b is not None:
                _store(m, "bias", b)
            m.use_temporary_parameter = False


def _store(module, name, value):
    """module.<name> <- value, in place when a tensor is already there (buffers keep their storage), else as a buffer."""
    cur = getattr(module, name, None)
    if isinstance(cur, torch.Tensor) and cur.shape == value.shape:
        cur.copy_(value)
        return
    if hasattr(module, name):
        delattr(module, name)
    module.register_buffer(name, value.detach().clone())
