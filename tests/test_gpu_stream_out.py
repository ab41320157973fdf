"""Outputs of 16 MiB and more per launch are written with non-temporal stores, smaller ones with plain stores.  Rows are
independent in every row-wise quantizer, so one matrix above the limit and its two halves below it must give the same
bits, forward and backward, in every family; the activation quantizer likewise for one large slab against the cached-
statistics kernels, which always store plainly."""
import pytest
import torch

from helpers import same
from miq import functional as F
from miq import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
O, I = 4608, 1024          # fp32: 18.9 MB in one launch, 9.4 MB per half
assert O * I * 4 >= (16 << 20) > O // 2 * I * 4


def _inputs(seed, dtype=torch.float32, rows=O):
    x = synth.weight(rows, I, seed).to(dtype).to(DEV)
    g = (synth.grad_like(x.cpu(), seed + 1) * 1e-3).to(dtype).to(DEV)
    up, low = (t.to(dtype).to(DEV) for t in synth.lwc_clips(rows, seed + 2))
    return x, g, up, low


def _halves(fn, x, g, up, low, *rowwise):
    """fn on the whole matrix and on its two halves; returns (whole, [first, second]) tuples of tensors."""
    h = x.shape[0] // 2
    cut = lambda t, a, b: None if t is None else t[a:b].contiguous()
    whole = fn(x, g, up, low, *rowwise)
    parts = [fn(cut(x, a, b), cut(g, a, b), cut(up, a, b), cut(low, a, b), *(cut(r, a, b) for r in rowwise))
             for a, b in ((0, h), (h, x.shape[0]))]
    return whole, parts


def _check(whole, parts):
    for k, w in enumerate(whole):
        if w is None:
            continue
        joined = torch.cat([p[k] for p in parts])
        assert same(w.cpu(), joined.cpu()), f"output {k}"


def test_uniform_rows():
    x, g, up, low = _inputs(11)

    def fn(x, g, up, low):
        out, scale, rzp, _ = F.rowquant_uniform_fwd(x, 4, False, up, low)
        return (out, scale, rzp) + tuple(F.rowquant_uniform_bwd(g, x, 4, False, up, low))
    _check(*_halves(fn, x, g, up, low))


def test_uniform_rows_fp16_and_fixed_mixed_bits():
    x, g, up, low = _inputs(12, torch.float16, rows=2 * O)  # fp16: twice the rows for the same bytes
    bits = torch.tensor([2.0, 3.0, 4.0], device=DEV)[torch.arange(2 * O, device=DEV) % 3].contiguous()

    def fn(x, g, up, low, bits):
        out, scale, rzp, _ = F.rowquant_uniform_fwd(x, 4, False, up, low, row_bits=bits, clamp_asym=True)
        return (out, scale, rzp) + tuple(F.rowquant_uniform_bwd(g, x, 4, False, up, low, row_bits=bits, clamp_asym=True))
    _check(*_halves(fn, x, g, up, low, bits))


@pytest.mark.parametrize("which", ["exp2", "exp1"])
def test_exponent_rows(which):
    x, g, up, low = _inputs(13)

    def fn(x, g, up, low):
        if which == "exp2":
            return tuple(F.rowquant_exp2_fwd(x, 3, up, low)) + tuple(F.rowquant_exp2_bwd(g, x, 3, up, low))
        return tuple(F.rowquant_exp1_fwd(x, 3, False, up, low)) + tuple(F.rowquant_exp1_bwd(g, x, 3, False, up, low))
    _check(*_halves(fn, x, g, up, low))


@pytest.mark.parametrize("hard", [False, True])
def test_learnable_mixed_bit_rows(hard):
    x, g, up, low = _inputs(14)
    nb = (3.5 + 0.25 * torch.rand(O, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))).contiguous()
    if hard:
        nb = nb.round()

    def fn(x, g, up, low, nb):
        return (F.rowquant_mix_fwd(x, nb, False, up, low, hard=hard),) + tuple(F.rowquant_mix_bwd(g, x, nb, False, up, low, hard=hard))
    _check(*_halves(fn, x, g, up, low, nb))


def test_batched_rows_above_the_limit_equal_single_matrices_below_it():
    xs, gs, ups, lows = zip(*[_inputs(20 + 3 * k, rows=O // 2) for k in range(3)])  # 3 x 9.4 MB in one launch
    outs, scales, rzps = F.rowquant_uniform_fwd_batch(list(xs), 4, False, list(ups), list(lows))
    dxs, dus, dls = F.rowquant_uniform_bwd_batch(list(gs), list(xs), 4, False, list(ups), list(lows))
    for k in range(3):
        out, scale, rzp, _ = F.rowquant_uniform_fwd(xs[k], 4, False, ups[k], lows[k])
        dx, du, dl = F.rowquant_uniform_bwd(gs[k], xs[k], 4, False, ups[k], lows[k])
        for a, b in ((outs[k], out), (scales[k], scale), (rzps[k], rzp), (dxs[k], dx), (dus[k], du), (dls[k], dl)):
            assert same(a.cpu(), b.cpu()), k


def test_activation_slab_above_the_limit():
    R, C, counts = 2048, 4096, [128] * 32   # fp32: 33.5 MB
    x = synth.activation((R, C), 31, torch.float32).to(DEV)
    layout = F.ClusterLayout.get(counts, C, DEV)
    outs, codes, vecs = F.act_fresh_quant_fwd([x], 4, layout, want_codes=True)
    ref, ref_codes, st = F.act_calib_quant_fwd(x, 4, "minmax", layout, want_codes=True)
    assert torch.equal(outs[0], ref) and torch.equal(codes[0], ref_codes) and torch.equal(vecs[0][2], st["scale"])


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_gather_above_the_limit(dtype):
    """The channel gather streams its output from 16 MiB per launch on: a tensor above the limit, and a batch of two whose
    sum is above it, equal torch's index_select."""
    rows, C = 4608, 2048 if dtype == torch.float16 else 1024   # 18.9 MB
    x = synth.activation((rows, C), 31, dtype).to(DEV)
    idx = synth.permutation(C, 32).to(torch.int32).to(DEV)
    assert x.numel() * x.element_size() >= (16 << 20)
    assert same(F.gather_lastdim(x, idx), x.index_select(-1, idx.long()))
    halves = [x[: rows // 2].contiguous(), x[rows // 2:].contiguous()]
    idx2 = synth.permutation(C, 33).to(torch.int32).to(DEV)
    outs = F.gather_lastdim_batch(halves, [idx, idx2])
    assert same(outs[0], halves[0].index_select(-1, idx.long())) and same(outs[1], halves[1].index_select(-1, idx2.long()))
