"""The fused calibration loss (miq_mse_fwd / miq_mse_bwd) against torch.nn.MSELoss under fp16 autocast, the way the
calibration loop of the reference evaluates it (quantize/omniquant.py:339, 511-513): loss to fp32 summation accuracy,
gradients bit for bit, for either operand."""
import pytest
import torch

from miq import functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("shape", [(1, 2048, 4096), (2, 33, 100), (1, 7), (3, 5, 8)])
@pytest.mark.parametrize("which", ["second", "first", "both"])
def test_loss_and_gradients_match_mseloss_under_autocast(shape, which):
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    a = torch.randn(*shape, generator=g, device=DEV).half()
    b = (a.float() + 0.05 * torch.randn(*shape, generator=g, device=DEV)).half()
    scale = torch.tensor(65536.0, device=DEV)
    res = []
    for fn in (F.mse_loss, torch.nn.MSELoss()):
        x = a.clone().requires_grad_(which in ("first", "both"))
        y = b.clone().requires_grad_(which in ("second", "both"))
        with torch.autocast("cuda", dtype=torch.float16):
            if fn is F.mse_loss:
                assert F.mse_supported(x, y)
            loss = fn(x, y)
        assert loss.dtype == torch.float32 and loss.dim() == 0
        grads = torch.autograd.grad(loss * scale, [t for t in (x, y) if t.requires_grad])
        res.append((loss.detach(), grads))
    (l0, g0), (l1, g1) = res
    assert abs(l0.item() - l1.item()) <= 1e-5 * abs(l1.item())
    for p, q in zip(g0, g1):
        assert p.dtype == q.dtype == torch.float16
        assert torch.equal(p, q)


def test_other_inputs_take_the_torch_path():
    a = torch.randn(4, 8, device=DEV)
    b = torch.randn(4, 8, device=DEV)
    assert not F.mse_supported(a, b)
    with torch.autocast("cuda", dtype=torch.float16):
        assert not F.mse_supported(a, b)                       # fp32 operands
        assert F.mse_supported(a.half(), b.half())
        assert torch.equal(F.mse_loss(a, b), torch.nn.functional.mse_loss(a, b))
    assert not F.mse_supported(a.half(), b.half())             # no autocast: fp16 arithmetic in the reference
