"""Fused reorder-gather + LayerNorm forward (miq_gather_layernorm_fwd) against
index_select + torch.nn.functional.layer_norm evaluated in fp32 on the CPU.

Floating-point contract: the kernel reduces mean / variance in fp32 (two passes over registers);
PyTorch reduces in another order, so fp32 outputs agree to 2e-6 of the row's scale and fp16 outputs
to one fp16 ulp of the fp32 result.  Saved mean / rstd: 1e-6 relative.  Backward (ATen's LayerNorm
backward on those statistics + inverse gather): 1e-5 / 2e-3 against autograd through the unfused ops.
"""
import pytest
import torch

from helpers import DT, rel_err, same
from miq import functional as F
from miq import synth
from miq.omni_norm import OmniLayerNorm

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def ref_ln(x, idx, w, b, eps):
    xg = torch.index_select(x.float(), -1, idx) if idx is not None else x.float()
    return torch.nn.functional.layer_norm(xg, (x.shape[-1],), None if w is None else w.float(), None if b is None else b.float(), eps)


@pytest.mark.parametrize("shape,xdt,wdt,odt,use_idx", [
    ((2, 37, 64), "f32", "f32", "f32", True),
    ((1, 2048, 4096), "f16", "f16", "f16", True),
    ((1, 2048, 4096), "f16", "f32", "f32", True),      # autocast: fp16 activations normalised in fp32
    ((2048, 4096), "f32", "f32", "f32", True),
    ((300, 768), "f32", "f32", "f32", True),
    ((129, 3072), "f16", "f16", "f16", True),
    ((64, 11008), "f32", "f32", "f32", True),
    ((16, 28672), "f16", "f16", "f16", True),
    ((5, 8), "f32", "f32", "f32", False),
    ((77, 5120), "f32", "f32", "f16", False),
])
def test_forward_matches_gather_then_layernorm(shape, xdt, wdt, odt, use_idx):
    C = shape[-1]
    x = synth.activation(shape, 60, DT[xdt])
    g = synth.gen(61)
    w = (torch.rand(C, generator=g) + 0.5).to(DT[wdt])
    b = (torch.randn(C, generator=g) * 0.1).to(DT[wdt])
    idx = synth.permutation(C, 62) if use_idx else None
    eps = 1e-5
    y, mean, rstd = F.gather_layernorm_fwd(x.to(DEV), None if idx is None else idx.to(torch.int32).to(DEV), w.to(DEV), b.to(DEV),
                                           eps, DT[odt])
    ref = ref_ln(x, idx, w, b, eps)
    assert y.dtype == DT[odt] and y.shape == x.shape
    if odt == "f32":
        assert (y.cpu() - ref).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item())
    else:
        assert rel_err(y.cpu().float(), ref) <= 1e-3
        assert ((y.cpu().float() - ref).abs() <= 1.01 * torch.maximum(ref.abs(), torch.tensor(6e-5)) * 2 ** -10).all()
    xf = x.float()
    assert rel_err(mean.cpu().reshape(-1), xf.mean(-1).reshape(-1)) <= 1e-5
    assert rel_err(rstd.cpu().reshape(-1), (xf.var(-1, unbiased=False) + eps).rsqrt().reshape(-1)) <= 1e-6
    # without weight / bias
    y0, _, _ = F.gather_layernorm_fwd(x.to(DEV), None if idx is None else idx.to(torch.int32).to(DEV), None, None, eps, DT[odt])
    assert rel_err(y0.cpu().float(), ref_ln(x, idx, None, None, eps)) <= (2e-6 if odt == "f32" else 1e-3)


@pytest.mark.parametrize("dt,amp", [("f32", False), ("f16", False), ("f16", True)])
def test_module_fused_equals_unfused_incl_backward(dt, amp):
    C, shape = 768, (2, 50, 768)
    ln = torch.nn.LayerNorm(C)
    g = synth.gen(70)
    with torch.no_grad():
        ln.weight.copy_(torch.rand(C, generator=g) + 0.5)
        ln.bias.copy_(torch.randn(C, generator=g) * 0.1)
    wdt = torch.float32 if (amp or dt == "f32") else torch.float16
    idx = synth.permutation(C, 71)
    x = synth.activation(shape, 72, DT[dt]).to(DEV)
    gy = synth.grad_like(torch.empty(shape), 73).to(DEV)
    outs = []
    for fused in (True, False):
        m = OmniLayerNorm(ln).to(DEV).to(wdt)
        m.register_buffer("reorder_index", idx.to(DEV))
        m.fused_gather = fused
        # LET-style temporary parameters that require grad
        m.temp_weight = m.weight.detach().clone().requires_grad_(True)
        m.temp_bias = m.bias.detach().clone().requires_grad_(True)
        m.use_temporary_parameter = True
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
            y = m(xi)
        grads = torch.autograd.grad((y.float() * gy).sum(), [xi, m.temp_weight, m.temp_bias])
        outs.append((y.detach(), grads))
    (y1, g1), (y0, g0) = outs
    assert y1.dtype == y0.dtype and y1.shape == y0.shape
    tol = 2e-6 if y0.dtype == torch.float32 else 1e-3
    assert rel_err(y1.float().cpu(), y0.float().cpu()) <= tol
    gt = 1e-5 if dt == "f32" else 2e-3
    for a, b, n in zip(g1, g0, ("dx", "dw", "db")):
        assert a.dtype == b.dtype and rel_err(a.float().cpu(), b.float().cpu()) <= gt, n
