"""Batched row quantizer (several weight matrices of one width in one launch): identical, bit for
bit, to one call per matrix — the batch only changes which launch a row belongs to."""
import pytest
import torch

from helpers import DT, same
from miq import _lib as L
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
        pairs = [synth.lwc_clips(o, 100 + i, dtype) for i, o in enumerate(rows)]
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
    pairs = [synth.lwc_clips(o, 130 + i) for i, o in enumerate((64, 96))]
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


@pytest.mark.parametrize("dt", ["f32", "f16"])
@pytest.mark.parametrize("lwc", [True, False])
def test_exp2_batch_equals_single_matrix_calls(dt, lwc):
    """The exponent (v2) quantizer over several matrices of one width in one launch each way: the bits of the
    single-matrix entry points, forward and backward, with and without clipping factors and dx."""
    dtype = DT[dt]
    I = 1024
    rows = [64, 1, 130, 257]
    xs = [synth.weight(o, I, 70 + i, dtype).to(DEV) for i, o in enumerate(rows)]
    gs = [synth.grad_like(x.cpu(), 80 + i).to(DEV) for i, x in enumerate(xs)]
    ups = lows = None
    if lwc:
        fac = [synth.lwc_clips(o, 90 + i, dtype) for i, o in enumerate(rows)]
        ups, lows = [f[0].to(DEV) for f in fac], [f[1].to(DEV) for f in fac]
    outs, scales = F.rowquant_exp2_fwd_batch(xs, 3, ups, lows)
    for want_dx in (True, False):
        dxs, d_ups, d_lows = F.rowquant_exp2_bwd_batch(gs, xs, 3, ups, lows, want_dx=want_dx)
        for i, x in enumerate(xs):
            u, l = (ups[i], lows[i]) if lwc else (None, None)
            o1, s1 = F.rowquant_exp2_fwd(x, 3, u, l)
            assert same(outs[i].cpu(), o1.cpu()) and same(scales[i].cpu(), s1.cpu())
            dx1, du1, dl1 = F.rowquant_exp2_bwd(gs[i], x, 3, u, l, want_dx=want_dx)
            if want_dx:
                assert same(dxs[i].cpu(), dx1.cpu())
            else:
                assert dxs is None
            if lwc:
                assert same(d_ups[i].cpu(), du1.cpu()) and same(d_lows[i].cpu(), dl1.cpu())
    with pytest.raises(L.MiqError):
        F.rowquant_exp2_fwd_batch([xs[0], xs[1][:, :512].contiguous()], 3)


@pytest.mark.parametrize("dt", ["f32", "f16"])
@pytest.mark.parametrize("lwc", [True, False])
@pytest.mark.parametrize("sym", [True, False])
def test_exp1_batch_equals_single_matrix_calls(dt, lwc, sym):
    """The exponent (v1) quantizer over several matrices of one width in one launch each way: the bits of the
    single-matrix entry points (output, scale, midpoint; dx and the clip-factor gradients), symmetric and around the
    row midpoint, with and without clipping factors and dx, on a narrow and on a wide (two-wave team) tile."""
    dtype = DT[dt]
    for I, rows in ((1024, [64, 1, 130, 257]), (11008, [3, 40])):
        xs = [synth.weight(o, I, 170 + i, dtype).to(DEV) for i, o in enumerate(rows)]
        gs = [synth.grad_like(x.cpu(), 180 + i).to(DEV) for i, x in enumerate(xs)]
        ups = lows = None
        if lwc:
            fac = [synth.lwc_clips(o, 190 + i, dtype) for i, o in enumerate(rows)]
            ups, lows = [f[0].to(DEV) for f in fac], [f[1].to(DEV) for f in fac]
        outs, scales, mids = F.rowquant_exp1_fwd_batch(xs, 3, sym, ups, lows)
        assert (mids is None) == sym
        for want_dx in (True, False):
            dxs, d_ups, d_lows = F.rowquant_exp1_bwd_batch(gs, xs, 3, sym, ups, lows, want_dx=want_dx)
            for i, x in enumerate(xs):
                u, l = (ups[i], lows[i]) if lwc else (None, None)
                o1, s1, m1 = F.rowquant_exp1_fwd(x, 3, sym, u, l)
                assert same(outs[i].cpu(), o1.cpu()) and same(scales[i].cpu(), s1.cpu())
                if not sym:
                    assert same(mids[i].cpu(), m1.cpu())
                dx1, du1, dl1 = F.rowquant_exp1_bwd(gs[i], x, 3, sym, u, l, want_dx=want_dx)
                if want_dx:
                    assert same(dxs[i].cpu(), dx1.cpu())
                else:
                    assert dxs is None
                if lwc:
                    assert same(d_ups[i].cpu(), du1.cpu()) and same(d_lows[i].cpu(), dl1.cpu())
    with pytest.raises(L.MiqError):
        F.rowquant_exp1_fwd_batch([xs[0], xs[1][:, :512].contiguous()], 3, sym)


def test_block_with_batched_exponent_weights_trains_like_one_launch_per_matrix():
    """An OPT-125m shaped block with the exponent (v2) weight quantizer: the calibration step that quantizes the
    weight matrices of one width in one launch each way ends at the same parameters, bit for bit, as the step that
    launches per matrix (the batch only changes which launch a row belongs to)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import bench_block
    from miq import blocks
    from miq.calibrate import CalibConfig, calibrate_layer

    def one(batched):
        blocks._BlockBase.batch_weights = batched
        try:
            torch.manual_seed(0)
            args = bench_block.quant_args(wbits=3, abits=8, lwc=True, a_dynamic_method="per_tensor", a_dynamic=False,
                                          metric="ema_minmax", let=True, weight_exp_quant=True, w_symmetric=True)
            qlayer, _ = bench_block.synthetic_block("opt-125m", DEV, args)
            g = torch.Generator(device=DEV).manual_seed(5)
            inps = torch.randn(3, 128, 768, generator=g, device=DEV)
            mask = torch.full((128, 128), torch.finfo(torch.float32).min, device=DEV).triu(1).view(1, 1, 128, 128)
            cfg = CalibConfig(epochs=2, nsamples=3, batch_size=1, let=True, lwc=True, deactive_amp=True, reorder="")
            prefix = "model.decoder.layers.0"
            sc = {f"{prefix}.{n}": torch.rand(768, generator=g, device=DEV) * 3 + 0.2 for n in ("self_attn.q_proj", "self_attn.out_proj", "fc1")}
            sh = {f"{prefix}.{n}": torch.randn(768, generator=g, device=DEV) * 0.1 for n in ("self_attn.q_proj", "self_attn.out_proj", "fc1")}
            res = calibrate_layer(qlayer, inps, inps.clone(), mask, cfg, sc, sh, prefix, is_llama=False)
            return {k: v.detach().clone() for k, v in qlayer.named_parameters()}, res
        finally:
            blocks._BlockBase.batch_weights = True

    calls = []
    real = F.rowquant_exp2_fwd_batch
    F.rowquant_exp2_fwd_batch = lambda xs, *a, **k: (calls.append(len(xs)), real(xs, *a, **k))[1]
    try:
        pa, ra = one(True)
        assert calls and max(calls) >= 4, calls   # q, k, v, out_proj and fc1 share the 768-wide launch
        n = len(calls)
        pb, rb = one(False)
        assert len(calls) == n                    # per-matrix launches only
    finally:
        F.rowquant_exp2_fwd_batch = real
    assert pa.keys() == pb.keys() and len(pa) > 0
    for k in pa:
        assert same(pa[k].cpu(), pb[k].cpu()), k


@pytest.mark.parametrize("sym", [False, True])
@pytest.mark.parametrize("hard", [False, True])
@pytest.mark.parametrize("I,rows,lwc", [
    (1024, [64, 1, 130, 257], True),
    (5120, [40, 3], True),       # Llama-2-13B width
    (13824, [5, 9], False),      # the 512-lane tile
    (100, [7, 0, 9], True),      # unaligned rows, an empty matrix
])
def test_mix_batch_equals_single_matrix_calls(sym, hard, I, rows, lwc):
    """The learnable mixed-bit quantizer over several matrices of one width in one launch each way: the bits of the
    single-matrix entry points (output, dx, d_up / d_low, d n_bits), soft and hard phase."""
    xs = [synth.weight(o, I, 170 + i).to(DEV) for i, o in enumerate(rows)]
    gs = [synth.grad_like(x.cpu(), 180 + i).to(DEV) for i, x in enumerate(xs)]
    bits = [synth.mix_bits(o, 3.0, 190 + i).to(DEV) for i, o in enumerate(rows)]
    ups = lows = None
    if lwc:
        fac = [synth.lwc_clips(o, 200 + i) for i, o in enumerate(rows)]
        ups, lows = [f[0].to(DEV) for f in fac], [f[1].to(DEV) for f in fac]
    outs = F.rowquant_mix_fwd_batch(xs, bits, sym, ups, lows, hard=hard, threshold=0.1)
    for want_dx in (True, False):
        dxs, d_ups, d_lows, d_bits = F.rowquant_mix_bwd_batch(gs, xs, bits, sym, ups, lows, hard=hard, threshold=0.1,
                                                              want_dx=want_dx)
        for i, x in enumerate(xs):
            if x.shape[0] == 0:
                continue
            u, l = (ups[i], lows[i]) if lwc else (None, None)
            o1 = F.rowquant_mix_fwd(x, bits[i], sym, u, l, hard=hard, threshold=0.1)
            assert same(outs[i].cpu(), o1.cpu()), i
            dx1, du1, dl1, dn1 = F.rowquant_mix_bwd(gs[i], x, bits[i], sym, u, l, hard=hard, threshold=0.1, want_dx=want_dx)
            if want_dx:
                assert same(dxs[i].cpu(), dx1.cpu()), i
            else:
                assert dxs is None
            assert same(d_bits[i].cpu(), dn1.cpu()), i
            if lwc:
                assert same(d_ups[i].cpu(), du1.cpu()) and same(d_lows[i].cpu(), dl1.cpu()), i


def test_mix_batch_wrappers_refuse_bad_tables():
    xs = [synth.weight(r, 64, 3 + i).to(DEV) for i, r in enumerate((8, 16))]
    bits = [synth.mix_bits(r, 3.0, 5 + i).to(DEV) for i, r in enumerate((8, 16))]
    gs = [torch.randn_like(x) for x in xs]
    F.rowquant_mix_fwd_batch(xs, bits, False)
    bad = [
        lambda: F.rowquant_mix_fwd_batch(xs, bits[:1], False),
        lambda: F.rowquant_mix_fwd_batch(xs, [bits[0], bits[0]], False),             # 8 bit-widths for 16 rows
        lambda: F.rowquant_mix_fwd_batch([x.half() for x in xs], bits, False),
        lambda: F.rowquant_mix_bwd_batch(gs[:1], xs, bits, False),
        lambda: F.rowquant_mix_bwd_batch(gs, xs, [b.double() for b in bits], False),
    ]
    for fn in bad:
        with pytest.raises(L.MiqError):
            fn()


def test_mix_batch_autograd_function_equals_per_matrix_function():
    I = 256
    rows = (64, 96)
    xs = [synth.weight(o, I, 220 + i).to(DEV).requires_grad_(True) for i, o in enumerate(rows)]
    bits = [synth.mix_bits(o, 3.0, 230 + i).to(DEV).requires_grad_(True) for i, o in enumerate(rows)]
    fac = [synth.lwc_clips(o, 240 + i) for i, o in enumerate(rows)]
    ups = [f[0].to(DEV).requires_grad_(True) for f in fac]
    lows = [f[1].to(DEV).requires_grad_(True) for f in fac]
    flat = []
    for x, b, u, l in zip(xs, bits, ups, lows):
        flat += [x, b, u, l]
    outs = F.MixRowFQBatch.apply(False, False, 0.0, True, *flat)
    got = torch.autograd.grad(sum((o * o).sum() for o in outs), flat)
    ref = []
    for x, b, u, l in zip(xs, bits, ups, lows):
        o = F.MixRowFQ.apply(x, b, u, l, False, 0, False, 0.0)
        ref += list(torch.autograd.grad((o * o).sum(), [x, b, u, l]))
    for a, b in zip(got, ref):
        assert a.shape == b.shape and torch.equal(a, b)


@pytest.mark.parametrize("dt,I,rows,lwc,sym", [
    ("f32", 5120, [40, 3, 17], True, False),
    ("f32", 13824, [5, 9], True, False),
    ("f16", 1024, [64, 0, 130], True, True),
    ("f32", 100, [7, 9], False, False),
])
def test_mix2_batch_equals_single_matrix_calls(dt, I, rows, lwc, sym):
    """The fixed mixed-bit quantizer (per-row integer bit-widths, clamped scale) over several matrices of one width in
    one launch each way: the bits of the single-matrix entry points."""
    dtype = DT[dt]
    xs = [synth.weight(o, I, 270 + i, dtype).to(DEV) for i, o in enumerate(rows)]
    gs = [synth.grad_like(x.cpu(), 280 + i).to(DEV) for i, x in enumerate(xs)]
    bits = [synth.mix2_bits(o, 290 + i).to(DEV) for i, o in enumerate(rows)]
    ups = lows = None
    if lwc:
        fac = [synth.lwc_clips(o, 300 + i, dtype) for i, o in enumerate(rows)]
        ups, lows = [f[0].to(DEV) for f in fac], [f[1].to(DEV) for f in fac]
    outs, scales, rzps = F.rowquant_mix2_fwd_batch(xs, bits, sym, ups, lows)
    for want_dx in (True, False):
        dxs, d_ups, d_lows = F.rowquant_mix2_bwd_batch(gs, xs, bits, sym, ups, lows, want_dx=want_dx)
        for i, x in enumerate(xs):
            if x.shape[0] == 0:
                continue
            u, l = (ups[i], lows[i]) if lwc else (None, None)
            o, s, z, _c = F.rowquant_uniform_fwd(x, 4, sym, u, l, 0, bits[i], True)
            assert same(outs[i].cpu(), o.cpu()) and same(scales[i].cpu(), s.cpu()) and same(rzps[i].cpu(), z.cpu()), i
            dx, du, dl = F.rowquant_uniform_bwd(gs[i], x, 4, sym, u, l, 0, bits[i], True, want_dx=want_dx)
            if want_dx:
                assert same(dxs[i].cpu(), dx.cpu()), i
            else:
                assert dxs is None
            if lwc:
                assert same(d_ups[i].cpu(), du.cpu()) and same(d_lows[i].cpu(), dl.cpu()), i
    with pytest.raises(L.MiqError):
        F.rowquant_mix2_fwd_batch(xs, bits[:1], sym, ups, lows)


@pytest.mark.parametrize("kind", ["mix", "mix2"])
def test_block_with_batched_mixed_bit_weights_trains_like_one_launch_per_matrix(kind):
    """A small OPT-shaped block whose weights go through the learnable (soft and hard phase, with the size loss) or
    the fixed mixed-bit quantizer: the step that quantizes the weight matrices of one width in one launch each way
    ends at the same parameters, bit for bit, as the step that launches per matrix."""
    from miq import blocks
    from miq.blocks import QuantOPTBlock, quant_args
    from miq.calibrate import CalibConfig, calibrate_layer
    H, FF, heads, T = 128, 512, 4, 64

    def one(batched):
        blocks._BlockBase.batch_weights = batched
        try:
            torch.manual_seed(0)
            g = torch.Generator(device=DEV).manual_seed(7)
            args = quant_args(wbits=3, abits=16, lwc=True, a_dynamic_method="per_tensor", a_dynamic=False, metric="ema_minmax",
                              let=False, weight_mix_quant=(kind == "mix"), weight_mix2_quant=(kind == "mix2"))
            w = lambda o, i: torch.randn(o, i, generator=g, device=DEV, dtype=torch.float16) * 0.05
            b = lambda o: torch.randn(o, generator=g, device=DEV, dtype=torch.float16) * 0.02
            one_ = torch.ones(H, device=DEV, dtype=torch.float16)
            weights = {"q": w(H, H), "k": w(H, H), "v": w(H, H), "o": w(H, H), "fc1": w(FF, H), "fc2": w(H, FF),
                       "q_b": b(H), "k_b": b(H), "v_b": b(H), "o_b": b(H), "fc1_b": b(FF), "fc2_b": b(H),
                       "ln1_w": one_, "ln1_b": b(H), "ln2_w": one_.clone(), "ln2_b": b(H)}
            nbits = None
            if kind == "mix2":
                rows = {"q": H, "k": H, "v": H, "o": H, "fc1": FF, "fc2": H}
                nbits = {n: synth.mix2_bits(r, 310 + i).to(DEV) for i, (n, r) in enumerate(rows.items())}
            qlayer = QuantOPTBlock(weights, heads, args, nbits=nbits).to(DEV)
            inps = torch.randn(4, T, H, generator=g, device=DEV)
            mask = torch.full((T, T), torch.finfo(torch.float32).min, device=DEV).triu(1).view(1, 1, T, T)
            cfg = CalibConfig(epochs=4 if kind == "mix" else 2, nsamples=4, batch_size=1, let=False, lwc=True, deactive_amp=True,
                              reorder="", weight_mix_quant=(kind == "mix"), loss_size_coef=0.1)
            calibrate_layer(qlayer, inps, inps.clone(), mask, cfg, None, None, "model.decoder.layers.0", is_llama=False)
            return {k: v.detach().clone() for k, v in qlayer.named_parameters()}
        finally:
            blocks._BlockBase.batch_weights = True

    name = "rowquant_mix_fwd_batch" if kind == "mix" else "rowquant_mix2_fwd_batch"
    calls = []
    real = getattr(F, name)
    setattr(F, name, lambda xs, *a, **k: (calls.append(len(xs)), real(xs, *a, **k))[1])
    try:
        pa = one(True)
        assert calls and max(calls) >= 4, calls   # q, k, v, out_proj and fc1 share the 128-wide launch
        n = len(calls)
        pb = one(False)
        assert len(calls) == n
    finally:
        setattr(F, name, real)
    assert pa.keys() == pb.keys() and len(pa) > 0
    for k in pa:
        assert same(pa[k].cpu(), pb[k].cpu()), k
