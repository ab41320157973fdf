"""miq_col_sum (bias gradient of QuantLinear's F.linear, reference quantize/int_linear.py:86) against an fp64 sum, and
the LinearBias autograd Function against torch.nn.functional.linear: same outputs and input / weight gradients (same
GEMMs), bias gradient to fp32 summation accuracy."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as TF

from miq import functional as F
from miq.int_linear import QuantLinear

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("R,C,dt", [
    (2048, 4096, torch.float16), (2048, 11008, torch.float16), (333, 768, torch.float32), (1, 64, torch.float16),
    (5, 7, torch.float32), (4097, 30, torch.float16), (64, 8, torch.float16), (100000, 12, torch.float32),
])
def test_col_sum_matches_fp64(R, C, dt):
    g = torch.Generator(device=DEV).manual_seed(R + C)
    x = (torch.randn(R, C, generator=g, device=DEV) + 0.25).to(dt)
    got = F.col_sum(x)
    want = x.double().sum(0)
    assert got.dtype == torch.float32 and got.shape == (C,)
    tol = 1e-6 * x.double().abs().sum(0) + 1e-6
    assert torch.all((got.double() - want).abs() <= tol)
    assert torch.equal(got, F.col_sum(x))  # fixed summation order
    assert torch.equal(F.col_sum(x, round_to=torch.float16), got.half().float())


def test_col_sum_rounded_to_half_overflows_to_inf():
    x = torch.full((64, 16), 4096.0, device=DEV, dtype=torch.float16)
    assert torch.all(F.col_sum(x) == 64 * 4096.0)
    assert torch.all(torch.isinf(F.col_sum(x, round_to=torch.float16)))


def test_col_sum_of_a_misaligned_view():
    base = torch.randn(64 * 16 + 1, device=DEV).half()
    x = base[1:].view(64, 16)  # 2-byte offset: the 16-byte path is not usable
    assert torch.allclose(F.col_sum(x), x.float().sum(0), rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize("amp", [False, True])
@pytest.mark.parametrize("shape", [(1, 256, 512, 384), (3, 40, 64, 200)])
def test_linear_bias_matches_functional_linear(amp, shape):
    B, T, I, O = shape
    g = torch.Generator(device=DEV).manual_seed(O)
    dt = torch.float16 if amp else torch.float32
    x = torch.randn(B, T, I, generator=g, device=DEV).to(dt)
    w = (torch.randn(O, I, generator=g, device=DEV) * 0.05)
    b = torch.randn(O, generator=g, device=DEV) * 0.1
    go = torch.randn(B, T, O, generator=g, device=DEV).to(dt)
    res = []
    for fn in (lambda *a: F.LinearBias.apply(*a), TF.linear):
        leaves = [t.clone().requires_grad_(True) for t in (x, w, b)]
        with torch.autocast("cuda", enabled=amp):
            assert F.linear_supported(*leaves)
            y = fn(*leaves)
        res.append((y, torch.autograd.grad(y, leaves, go)))
    (y0, g0), (y1, g1) = res
    assert y0.dtype == y1.dtype and torch.equal(y0, y1)
    for a, c in zip(g0, g1):
        assert a.dtype == c.dtype and a.shape == c.shape
    assert torch.equal(g0[0], g1[0]) and torch.equal(g0[1], g1[1])
    want = go.double().sum((0, 1))
    err = (g0[2].double() - want).abs().max().item()
    err_ref = (g1[2].double() - want).abs().max().item()
    assert err <= max(2 * err_ref, 1e-5 * want.abs().max().item())


def test_quant_linear_uses_it_only_for_a_trained_bias():
    lin = nn.Linear(64, 32).to(DEV)
    wq = {"n_bits": 4, "per_channel_axes": [0], "symmetric": False, "dynamic_method": "per_channel", "group_size": None,
          "lwc": True, "dynamic": True, "metric": "minmax", "target_bit": 4}
    aq = {"n_bits": 4, "per_channel_axes": [], "symmetric": False, "dynamic_method": "per_tensor", "dynamic": False,
          "metric": "minmax"}
    q = QuantLinear(lin, wq, aq).to(DEV)
    x = torch.randn(2, 5, 64, device=DEV)
    q.use_temporary_parameter = True
    q.temp_weight = q.weight
    q.temp_bias = q.bias.clone().requires_grad_(True)
    y = q(x)
    assert type(y.grad_fn).__name__ == "LinearBiasBackward"
    y.sum().backward()
    assert torch.allclose(q.temp_bias.grad, torch.full((32,), 10.0, device=DEV))
    q.fused_bias_grad = False
    assert type(q(x).grad_fn).__name__ != "LinearBiasBackward"
    q.fused_bias_grad = True
    q.temp_bias = q.bias  # a buffer: nothing to train
    assert q(x).grad_fn is None


def test_col_sum_is_linear_at_block_size():
    """[2048, 11008] (the widest bias gradient of a Llama-2-7B block): col_sum(a + b) = col_sum(a) + col_sum(b) and
    col_sum(2a) = 2 col_sum(a) to fp32 summation accuracy."""
    g = torch.Generator(device=DEV).manual_seed(8)
    a = torch.randn(2048, 11008, generator=g, device=DEV)
    b = torch.randn(2048, 11008, generator=g, device=DEV)
    sa, sb = F.col_sum(a), F.col_sum(b)
    assert torch.allclose(F.col_sum(a + b), sa + sb, rtol=1e-4, atol=2e-3)
    assert torch.equal(F.col_sum(2 * a), 2 * sa)
