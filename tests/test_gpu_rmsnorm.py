"""The fused OmniLlamaRMSNorm pass (miq_rmsnorm_fwd / _bwd) against the PyTorch chain of the reference module
(quantize/omni_norm.py:60-71): fp32 mean of squares, x * rsqrt, weight * x (+ bias), cast back.  Reduction order
differs from ATen's, so values agree to rounding of the result dtype; the weight / bias gradients are sums over all
rows and are compared against an fp64 evaluation of the same formula."""
import pytest
import torch
import torch.nn as nn

from miq import functional as F
from miq.omni_norm import OmniLlamaRMSNorm

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def chain(x, weight, bias, eps):
    var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
    h = x * torch.rsqrt(var + eps)
    return (weight * h + bias).to(x.dtype) if bias is not None else (weight * h).to(x.dtype)


CASES = [
    # leading shape, C, x dtype, vector dtype, bias
    ((2, 64), 4096, torch.float16, torch.float32, True),
    ((2, 64), 4096, torch.float16, torch.float16, False),
    ((3, 17), 768, torch.float32, torch.float32, True),
    ((1, 5), 5120, torch.float16, torch.float32, True),
    ((130,), 8192, torch.float32, torch.float32, False),
    ((7,), 64, torch.float16, torch.float16, True),
    ((1, 300), 11008, torch.float16, torch.float32, True),
    ((1,), 4, torch.float32, torch.float32, True),
]


@pytest.mark.parametrize("lead,C,dt,vdt,with_bias", CASES)
def test_forward_and_backward_match_the_pytorch_chain(lead, C, dt, vdt, with_bias):
    g = torch.Generator(device=DEV).manual_seed(C + len(lead))
    x = (torch.randn(*lead, C, generator=g, device=DEV) * 1.5).to(dt)
    w = (1 + 0.2 * torch.randn(C, generator=g, device=DEV)).to(vdt)
    b = (0.1 * torch.randn(C, generator=g, device=DEV)).to(vdt) if with_bias else None
    assert F.rmsnorm_supported(x, w, b)
    eps = 1e-6
    leaves = [t.clone().requires_grad_(True) for t in (x, w) + ((b,) if with_bias else ())]
    refs = [t.clone().requires_grad_(True) for t in (x, w) + ((b,) if with_bias else ())]
    out = F.RMSNorm.apply(leaves[0], leaves[1].float(), leaves[2].float() if with_bias else None, eps)
    ref = chain(refs[0], refs[1], refs[2] if with_bias else None, eps)
    assert out.dtype == ref.dtype and out.shape == ref.shape
    if dt == torch.float16:  # one fp16 rounding of nearly equal fp32 values
        assert (out.float() - ref.float()).abs().max().item() <= 2 ** -10 * max(1.0, ref.float().abs().max().item())
        assert (out != ref).float().mean().item() < 0.01
    else:
        assert torch.allclose(out, ref, rtol=2e-6, atol=2e-6)
    go = torch.randn(*lead, C, generator=g, device=DEV).to(dt)
    grads = torch.autograd.grad(out, leaves, go)
    # fp64 evaluation of the same formula
    r64 = [t.detach().double().requires_grad_(True) for t in (x, w) + ((b,) if with_bias else ())]
    var = r64[0].pow(2).mean(-1, keepdim=True)
    y64 = r64[1] * (r64[0] * torch.rsqrt(var + eps))
    if with_bias:
        y64 = y64 + r64[2]
    exact = torch.autograd.grad(y64, r64, go.double())
    ref_grads = torch.autograd.grad(ref, refs, go)
    for got, want, torch_got in zip(grads, exact, ref_grads):
        assert got.dtype == torch_got.dtype and got.shape == torch_got.shape
        scale = want.norm().clamp_min(1e-30)
        err = (got.double() - want).norm() / scale
        err_torch = (torch_got.double() - want).norm() / scale
        limit = 1e-3 if got.dtype == torch.float16 else 2e-6
        assert err.item() <= max(limit, 2 * err_torch.item()), (err.item(), err_torch.item())


def test_module_uses_the_kernel_and_keeps_the_chain_as_an_option():
    C = 512
    g = torch.Generator(device=DEV).manual_seed(3)
    ori = nn.Module()
    ori.weight = nn.Parameter((1 + 0.1 * torch.randn(C, generator=g, device=DEV)).half())
    m = OmniLlamaRMSNorm(ori, eps=1e-5)
    x = torch.randn(2, 9, C, generator=g, device=DEV).half()
    fused = m(x)
    m.fused = False
    plain = m(x)
    assert fused.dtype == plain.dtype == torch.float16
    assert (fused.float() - plain.float()).abs().max().item() <= 2 ** -10 * plain.float().abs().max().item()
    # temporary (smoothed) parameters receive gradients through the kernel
    m.fused = True
    m.use_temporary_parameter = True
    m.temp_weight = (ori.weight.detach().float() * 0.9).requires_grad_(True)
    m.temp_bias = torch.zeros(C, device=DEV).requires_grad_(True)
    y = m(x.float())
    y.square().mean().backward()
    assert m.temp_weight.grad is not None and m.temp_bias.grad is not None
    assert torch.isfinite(m.temp_weight.grad).all() and m.temp_weight.grad.abs().sum().item() > 0


def test_only_the_requested_gradients_are_computed():
    C = 256
    x = torch.randn(33, C, device=DEV)
    w = torch.ones(C, device=DEV, requires_grad=True)
    y = F.RMSNorm.apply(x, w, None, 1e-6)
    (dw,) = torch.autograd.grad(y, [w], torch.ones_like(y))
    u = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6)
    assert torch.allclose(dw, u.sum(0), rtol=1e-5, atol=1e-5)
    xr = x.clone().requires_grad_(True)
    y = F.RMSNorm.apply(xr, w.detach(), None, 1e-6)
    (dx,) = torch.autograd.grad(y, [xr], torch.ones_like(y))
    assert torch.isfinite(dx).all()


def test_unsupported_inputs_are_declined():
    assert not F.rmsnorm_supported(torch.zeros(4, 6, device=DEV), torch.ones(6, device=DEV), None)  # 6 % 4 != 0
    assert not F.rmsnorm_supported(torch.zeros(4, 8, device=DEV, dtype=torch.bfloat16), torch.ones(8, device=DEV), None)
    assert not F.rmsnorm_supported(torch.zeros(4, 8, device=DEV), None, None)
    assert not F.rmsnorm_supported(torch.zeros(0, 8, device=DEV), torch.ones(8, device=DEV), None)


def test_unit_rms_at_block_size():
    """[2048, 4096] fp16 hidden states with unit weight and no bias: every output row has root mean square
    sqrt(v / (v + eps)) with v the mean square of the input row (1 up to eps), to fp16 rounding, whatever the scale
    of the input row."""
    g = torch.Generator(device=DEV).manual_seed(5)
    x = (torch.randn(2048, 4096, generator=g, device=DEV) * torch.logspace(-2, 2, 2048, device=DEV)[:, None]).half()
    y = F.RMSNorm.apply(x, torch.ones(4096, device=DEV), None, 1e-6)
    rms = y.float().pow(2).mean(-1).sqrt()
    v = x.float().pow(2).mean(-1)
    assert (rms - torch.sqrt(v / (v + 1e-6))).abs().max().item() <= 1e-3
