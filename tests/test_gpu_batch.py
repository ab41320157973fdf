"""Batched row quantizer (several weight matrices of one width in one launch): identical, bit for
bit, to one call per matrix — the batch only changes which launch a row belongs to."""
import pytest
import torch

from helpers import DT
from miq import functional as F
from miq import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("dt,I,rows,lwc,sym,bits", [
    ("f32", 4096, [4096, 4096, 11008, 1], True, False, 4),
    ("f32", 768, [768, 768, 768, 768, 3072], True, False, 3),
    ("f16", 4096, [512, 0, 300], True, True, 4),
    ("f32", 100, [7, 9], False, False, 8),
    ("f16", 11008, [64, 33], False, False, 2),
])
def test_batch_equals_single_calls(dt, I, rows, lwc, sym, bits):
    dtype = DT[dt]
    xs = [synth.weight(o, I, 80 + i, dtype).to(DEV) for i, o in enumerate(rows)]
    gs = [synth.grad_like(x.cpu(), 90 + i).to(DEV) for i, x in enumerate(xs)]
    ups = lows = None
    if lwc:
        pairs = [synth.lwc_factors(o, 100 + i, dtype) for i, o in enumerate(rows)]
        ups = [p[0].to(DEV) for p in pairs]
        lows = [p[1].to(DEV) for p in pairs]
    outs, scales, rzps = F.rowquant_uniform_fwd_batch(xs, bits, sym, ups, lows)
    dxs, d_ups, d_lows = F.rowquant_uniform_bwd_batch(gs, xs, bits, sym, ups, lows)
    _n, only_lwc_up, only_lwc_low = F.rowquant_uniform_bwd_batch(gs, xs, bits, sym, ups, lows, want_dx=False)
    assert _n is None
    for i, x in enumerate(xs):
        u = ups[i] if lwc else None
        l = lows[i] if lwc else None
        o, s, z, _c = F.rowquant_uniform_fwd(x, bits, sym, u, l)
        dx, du, dl = F.rowquant_uniform_bwd(gs[i], x, bits, sym, u, l)
        assert torch.equal(outs[i], o) and torch.equal(scales[i], s) and torch.equal(rzps[i], z), i
        assert torch.equal(dxs[i], dx), i
        if lwc:
            assert torch.equal(d_ups[i], du) and torch.equal(d_lows[i], dl), i
            assert torch.equal(only_lwc_up[i], du) and torch.equal(only_lwc_low[i], dl), i


def test_batch_autograd_function():
    I = 256
    xs = [synth.weight(o, I, 120 + i).to(DEV).requires_grad_(True) for i, o in enumerate((64, 96))]
    pairs = [synth.lwc_factors(o, 130 + i) for i, o in enumerate((64, 96))]
    ups = [p[0].to(DEV).requires_grad_(True) for p in pairs]
    lows = [p[1].to(DEV).requires_grad_(True) for p in pairs]
    flat = []
    for x, u, l in zip(xs, ups, lows):
        flat += [x, u, l]
    res = F.UniformRowFQBatch.apply(4, False, True, *flat)
    outs = res[0::3]
    loss = sum((o * o).sum() for o in outs)
    got = torch.autograd.grad(loss, flat)
    ref = []
    for x, u, l in zip(xs, ups, lows):
        o, _s, _z = F.UniformRowFQ.apply(x, u, l, 4, False, 0, None, False)
        ref += list(torch.autograd.grad((o * o).sum(), [x, u, l]))
    for a, b in zip(got, ref):
        assert torch.equal(a, b)
    with pytest.raises(Exception):
        F.rowquant_uniform_fwd_batch([xs[0].detach(), torch.zeros(4, 128, device=DEV)], 4, False)


def test_batch_wrappers_refuse_what_the_kernels_cannot_check():
    """The batched kernels walk raw pointer tables: mis-sized factors, gradients of another shape or dtype, missing
    entries and more than 16 matrices must fail in the wrapper, not read or write out of bounds on the device."""
    from miq._lib import MiqError
    xs = [synth.weight(r, 64, 3 + i).to(DEV) for i, r in enumerate((8, 16))]
    ups = [torch.full((x.shape[0], 1), 4.0, device=DEV) for x in xs]
    lows = [torch.full((x.shape[0], 1), 4.0, device=DEV) for x in xs]
    gs = [torch.randn_like(x) for x in xs]
    F.rowquant_uniform_fwd_batch(xs, 4, False, ups, lows)
    F.rowquant_uniform_bwd_batch(gs, xs, 4, False, ups, lows)
    bad = [
        lambda: F.rowquant_uniform_fwd_batch(xs, 4, False, ups[:1], lows),             # one factor missing
        lambda: F.rowquant_uniform_fwd_batch(xs, 4, False, [ups[0], ups[0]], lows),    # 8 factors for 16 rows
        lambda: F.rowquant_uniform_fwd_batch(xs, 4, False, [u.half() for u in ups], lows),
        lambda: F.rowquant_uniform_fwd_batch(xs, 4, False, ups, None),
        lambda: F.rowquant_uniform_fwd_batch(xs * 9, 4, False),                        # 18 matrices
        lambda: F.rowquant_uniform_bwd_batch(gs[:1], xs, 4, False, ups, lows),
        lambda: F.rowquant_uniform_bwd_batch([gs[0], gs[0]], xs, 4, False, ups, lows),  # gradient of another shape
        lambda: F.rowquant_uniform_bwd_batch([g.half() for g in gs], xs, 4, False, ups, lows),
        lambda: F.rowquant_uniform_bwd_batch(gs * 9, xs * 9, 4, False),
    ]
    for i, fn in enumerate(bad):
        with pytest.raises(MiqError):
            fn()
