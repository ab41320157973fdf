"""miq_rotary against the PyTorch chain of transformers' apply_rotary_pos_emb as the reference's Llama attention
calls it (models/int_llama_layer.py:124-125): bit-exact forward and backward in fp16 and fp32, on the transposed
view of a projection output and on contiguous tensors."""
import pytest
import torch

from miq import functional as F
from miq.blocks import _Rotary, _rotate_half

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def chain(x, cos, sin):
    return x * cos + _rotate_half(x) * sin


def tables(B, T, D, dt, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    ang = torch.rand(B, 1, T, D, generator=g, device=DEV) * 6.28
    return ang.cos().to(dt).contiguous(), ang.sin().to(dt).contiguous()


@pytest.mark.parametrize("B,H,T,D,dt,tb,transposed", [
    (1, 32, 256, 128, torch.float16, 1, True),
    (2, 4, 33, 64, torch.float16, 2, True),
    (2, 4, 33, 64, torch.float32, 1, True),
    (1, 8, 40, 128, torch.float32, 1, False),
    (3, 2, 7, 16, torch.float16, 3, False),
    (1, 2, 5, 8, torch.float32, 1, True),
])
def test_bit_exact_with_the_pytorch_chain(B, H, T, D, dt, tb, transposed):
    g = torch.Generator(device=DEV).manual_seed(H * T + D)
    cos, sin = tables(tb, T, D, dt, 7)  # independent halves: more general than the duplicated-frequency tables
    base = torch.randn(B, T, H * D, generator=g, device=DEV).to(dt) * 3
    go = torch.randn(B, H, T, D, generator=g, device=DEV).to(dt)
    outs = []
    for fused in (True, False):
        leaf = base.clone().requires_grad_(True)
        x = leaf.view(B, T, H, D).transpose(1, 2)
        if not transposed:
            x = x.contiguous()
        assert F.rotary_supported(x, cos, sin)
        y = F.RotaryEmb.apply(x, cos, sin) if fused else chain(x, cos, sin)
        (dx,) = torch.autograd.grad(y, leaf, go)
        outs.append((y, dx))
    (y0, d0), (y1, d1) = outs
    assert y0.is_contiguous() and y0.dtype == dt
    assert torch.equal(y0, y1)
    assert torch.equal(d0, d1)


def test_rotary_module_uses_the_kernel_for_queries_and_grouped_keys():
    B, T, H, KV, D = 1, 64, 8, 2, 128
    rot = _Rotary(D)
    g = torch.Generator(device=DEV).manual_seed(1)
    q = torch.randn(B, T, H * D, generator=g, device=DEV).half().view(B, T, H, D).transpose(1, 2).requires_grad_(True)
    k = torch.randn(B, T, KV * D, generator=g, device=DEV).half().view(B, T, KV, D).transpose(1, 2)
    pos = torch.arange(T, device=DEV).unsqueeze(0)
    q1, k1 = rot(q, k, pos)
    assert type(q1.grad_fn).__name__ == "RotaryEmbBackward"
    rot.fused = False
    q0, k0 = rot(q, k, pos)
    assert torch.equal(q1, q0) and torch.equal(k1, k0)


def test_unsupported_layouts_are_declined():
    cos, sin = tables(1, 4, 16, torch.float16, 1)
    x = torch.zeros(1, 2, 4, 16, device=DEV, dtype=torch.float16)
    assert F.rotary_supported(x, cos, sin)
    assert not F.rotary_supported(x[..., ::2], cos, sin)                      # strided last dimension
    assert not F.rotary_supported(x.float(), cos, sin)                        # table dtype differs
    assert not F.rotary_supported(torch.zeros(1, 2, 4, 24, device=DEV, dtype=torch.float16), *tables(1, 4, 24, torch.float16, 1))


def test_round_trip_at_block_size():
    """Llama-2-7B query shape [1, 32, 2048, 128] fp16 with the module's own tables: the backward kernel applies the
    transposed (= inverse) rotation, so forward followed by backward returns the input to fp16 rounding, and the
    rotation keeps the norm of every head vector."""
    B, H, T, D = 1, 32, 2048, 128
    rot = _Rotary(D)
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn(B, T, H * D, generator=g, device=DEV).half().view(B, T, H, D).transpose(1, 2)
    cos, sin = rot.tables(torch.arange(T, device=DEV).unsqueeze(0), x.device, x.dtype)
    y = F.RotaryEmb.apply(x, cos, sin)
    back = torch.empty((B, T, H, D), dtype=x.dtype, device=DEV).transpose(1, 2)
    F._rotary_call(y, cos, sin, back, True)
    assert (back.float() - x.float()).abs().max().item() <= 1e-2
    assert torch.allclose(y.float().norm(dim=-1), x.float().norm(dim=-1), rtol=2e-3)
