"""The fused attention-score pass (miq_masked_softmax_fwd / _bwd) against the PyTorch chain it replaces in
the decoder blocks: scores / sqrt(d) + mask, floor at finfo.min, fp32 softmax, cast back.  The summation
order differs from ATen's, so the comparison is to rounding of the result dtype, forward and backward."""
import math

import pytest
import torch
import torch.nn.functional as TF

from miq import functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def chain(scores, mask, div):
    w = scores / div if div != 1.0 else scores
    if mask is not None:
        w = w + mask
        w = torch.max(w, torch.tensor(torch.finfo(w.dtype).min))  # the reference's form: ties pass half the gradient
    return TF.softmax(w, dim=-1, dtype=torch.float32).to(scores.dtype)


CASES = [
    # B, H, Q, K, scores dtype, mask dtype (None = no mask), mask batch, div
    (1, 32, 256, 256, torch.float16, torch.float32, 1, math.sqrt(128.0)),
    (2, 4, 128, 128, torch.float16, torch.float32, 2, math.sqrt(64.0)),
    (2, 4, 128, 128, torch.float32, torch.float32, 1, math.sqrt(64.0)),
    (1, 12, 64, 64, torch.float16, torch.float16, 1, 1.0),
    (1, 12, 96, 96, torch.float32, torch.float32, 1, 1.0),
    (3, 2, 40, 2048, torch.float16, torch.float32, 3, math.sqrt(128.0)),
    (1, 2, 24, 4096, torch.float32, None, 1, 1.0),
    (1, 2, 16, 16, torch.float16, None, 1, 2.0),
]


@pytest.mark.parametrize("B,H,Q,K,dt,mdt,mb,div", CASES)
def test_forward_and_backward_match_the_pytorch_chain(B, H, Q, K, dt, mdt, mb, div):
    g = torch.Generator(device=DEV).manual_seed(B * 1000 + K)
    scores = (torch.randn(B, H, Q, K, generator=g, device=DEV) * 4).to(dt)
    mask = None
    if mdt is not None:
        causal = torch.full((Q, K), torch.finfo(mdt).min, device=DEV, dtype=mdt).triu(1 + max(K - Q, 0))
        mask = causal.view(1, 1, Q, K).repeat(mb, 1, 1, 1)
        if mb > 1:  # per-sample padding on top of the causal part
            mask[1, :, :, : K // 8] = torch.finfo(mdt).min
    assert F.masked_softmax_supported(scores, mask)
    xs = scores.clone().requires_grad_(True)
    xr = scores.clone().requires_grad_(True)
    out = F.MaskedSoftmax.apply(xs, mask, div)
    ref = chain(xr, mask, div)
    assert out.dtype == ref.dtype and out.shape == ref.shape
    tol = 2e-3 if dt == torch.float16 else 2e-6
    assert (out.float() - ref.float()).abs().max().item() <= tol
    assert torch.equal(out == 0, ref == 0) or dt == torch.float16  # masked positions are exact zeros
    go = torch.randn(B, H, Q, K, generator=g, device=DEV).to(dt)
    (dx,) = torch.autograd.grad(out, xs, go)
    (dr,) = torch.autograd.grad(ref, xr, go)
    err = (dx.float() - dr.float()).norm() / dr.float().norm().clamp_min(1e-20)
    # both sides hold fp32 probabilities in the backward (for fp16 scores ours are rebuilt from the scores): what is left is
    # the rounding of the result to the scores' dtype where the two fp32 values straddle a rounding boundary
    assert err.item() <= (2e-4 if dt == torch.float16 else 1e-5), err.item()
    if mask is not None:
        # masked positions carry no gradient (probabilities that merely round to 0 in fp16 do, as in PyTorch; so do the
        # uniform probabilities of a row whose every key is masked)
        cut = (mask == torch.finfo(mdt).min).expand(B, 1, Q, K).expand(B, H, Q, K) & (out == 0)
        assert torch.all(dx[cut] == 0)


def test_fp16_backward_from_scores_beats_the_one_from_rounded_probabilities():
    """The reference keeps the fp32 softmax output for its backward.  The backward that rebuilds the fp32 probabilities
    (the one the autograd function uses for fp16 scores) is compared with the one that reads the rounded output, both
    against the gradient of the PyTorch chain itself, element for element."""
    from miq import _lib as L
    B, H, Q, K = 1, 8, 256, 512
    div = math.sqrt(128.0)
    g = torch.Generator(device=DEV).manual_seed(77)
    scores = (torch.randn(B, H, Q, K, generator=g, device=DEV) * 4).half()
    mask = torch.full((Q, K), torch.finfo(torch.float32).min, device=DEV).triu(1 + K - Q).view(1, 1, Q, K)
    go = torch.randn(B, H, Q, K, generator=g, device=DEV).half()
    xs = scores.clone().requires_grad_(True)
    out = F.MaskedSoftmax.apply(xs, mask, div)
    (dx,) = torch.autograd.grad(out, xs, go)
    dp = torch.empty_like(out)
    with L.on_device(out.device):
        L.call("miq_masked_softmax_bwd", L.ptr(out.detach()), L.ptr(go), L.ptr(dp), L.i64(B * H * Q), L.i64(K), L.f32(div),
               L.i32(L.dtype_code(out)), L.stream_of(out))
    xr = scores.clone().requires_grad_(True)
    (dr,) = torch.autograd.grad(chain(xr, mask, div), xr, go)
    same_scores = (dx == dr).float().mean().item()
    same_probs = (dp == dr).float().mean().item()
    print(f"bit-equal to the PyTorch chain's gradient: from scores {same_scores:.4f}, from rounded probabilities {same_probs:.4f}")
    assert same_scores >= 0.999, same_scores   # summation order of the softmax and of the dot product differ
    assert same_probs < same_scores


def test_rows_sum_to_one_at_block_size():
    """Llama-2-7B score shape: 32 heads x 2048 x 2048 in fp16 with an fp32 causal mask."""
    B, H, Q, K = 1, 32, 2048, 2048
    g = torch.Generator(device=DEV).manual_seed(5)
    scores = (torch.randn(B, H, Q, K, generator=g, device=DEV) * 8).half()
    mask = torch.full((Q, K), torch.finfo(torch.float32).min, device=DEV).triu(1).view(1, 1, Q, K)
    out = F.MaskedSoftmax.apply(scores, mask, math.sqrt(128.0))
    sums = out.float().sum(-1)
    assert (sums - 1).abs().max().item() <= 2e-3
    assert torch.all(out.triu(1) == 0)
    ref = chain(scores, mask, math.sqrt(128.0))
    assert (out.float() - ref.float()).abs().max().item() <= 2e-3


def test_unsupported_layouts_are_declined():
    s = torch.zeros(1, 2, 8, 12, device=DEV, dtype=torch.float16)  # 12 keys: not a multiple of the 8-element pack
    assert not F.masked_softmax_supported(s, None)
    s32 = torch.zeros(1, 2, 8, 16, device=DEV)
    assert not F.masked_softmax_supported(s32, torch.zeros(1, 1, 8, 16, device=DEV, dtype=torch.float16))
    assert not F.masked_softmax_supported(s32, torch.zeros(1, 2, 8, 16, device=DEV))
