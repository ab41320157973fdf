"""The fresh-statistics form of the calibrating activation quantizer (miq_act_fresh_quant_fwd / _bwd: statistics merged
by atomics into a slot table, up to 8 slabs per launch, fold + shares + sparse correction of the backward in one launch)
against the CPU oracle and against the four-kernel entry points it replaces in a training step.

Bars: forward bit-identical (outputs, integer codes, merged statistics, scale, zero points) to the oracle and to
miq_act_calib_quant_fwd; backward against miq_colquant_bwd: the direct path (every element away from its cluster's
extremes) bit-identical, the elements a share of the scale gradient is routed to equal to the rounding of that sum (the
two-launch form folds per-lane partials per cluster, the four-kernel form per-channel partials per channel: another
summation order; layouts that take the older one-launch fold stay bit-identical), and within the oracle tolerances of
test_gpu_act_fused.py.  Shapes: ragged row chunks, partial column tiles, clusters of one channel and
of more than 64 channels, unequal clusters (slot map), channels outside every cluster, C not a multiple of the pack
width, one cluster spanning everything, and the Llama-2 7B / 13B / 70B activation shapes.
"""
import pytest
import torch

from helpers import DT, rel_err, same, same_act_backward
from miq import _lib as L
from miq import functional as F
from miq import synth
from oracle import miq_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = [
    # shape, dtype, bits, cluster counts (None = one cluster over all channels)
    ((512, 1024), "f32", 4, [128] * 8),
    ((512, 1024), "f16", 4, [32] * 32),
    ((300, 2752), "f16", 8, [344] * 8),
    ((333, 1000), "f32", 4, [50] * 10 + [20] * 10 + [7, 93]),  # unequal clusters, 100 channels outside every cluster
    ((77, 520), "f16", 6, [1] * 20 + [400]),                  # single-channel clusters, 100 left over
    ((333, 1000), "f32", 4, [100, 300, 7, 293, 200]),
    ((130, 1000), "f32", 4, [100] * 9),                       # equal clusters, 100 channels left over
    ((5, 64), "f32", 3, [64]),
    ((1, 8), "f16", 4, [3, 5]),
    ((4100, 256), "f32", 4, None),
    ((2100, 2048), "f16", 4, None),                           # one cluster wider than the merged backward handles
    ((64, 100), "f16", 5, [5] * 20),                          # C not a multiple of the 8-element pack
    ((40, 1200), "f32", 4, [1200]),                           # one cluster of more than 1024 channels
    ((2100, 1024), "f16", 4, [512] * 2),                      # cluster slabs too large for the in-block mass-tie correction
]


def _old_path(xd, gd, bits, layout):
    out, codes, st = F.act_calib_quant_fwd(xd, bits, "minmax", layout, want_codes=True)
    dx = None
    if gd is not None:
        dx = F.colquant_bwd(gd, xd, st["scale"], st["rzp"], bits,
                            route=(st["chan_min"], st["chan_max"], st["cmin"], st["cmax"], layout))
    return out, codes, st, dx


@pytest.mark.parametrize("shape,dt,bits,counts", CASES)
def test_fresh_path_matches_oracle_and_four_kernel_path(shape, dt, bits, counts):
    dtype = DT[dt]
    R, C = shape
    layout = F.ClusterLayout.get(counts, C, DEV)
    kw = dict(method="per_cluster" if counts else "per_tensor", metric="minmax", cluster_dim=1 if counts else None,
              cluster_counts=counts)
    # three calls in a row on one statistics table: the third has a SMALLER range (stale maxima must not survive)
    # and mass ties at its minimum
    for call, (seed, relu, gain) in enumerate(((310, False, 1.0), (311, False, 2.0), (312, True, 0.25))):
        x = synth.activation(shape, seed, dtype) * gain
        x = (torch.relu(x) if relu else x).to(dtype)
        g = synth.grad_like(x, seed + 50)
        xd, gd = x.to(DEV), g.to(DEV)
        outs, codes, vecs = F.act_fresh_quant_fwd([xd], bits, layout, want_codes=True)
        st = F.stats_of(vecs[0])
        ref, ref_codes = O.act_forward(x, (ost := O.ActState()), bits, "calibration", **kw)
        vec = lambda t: t.reshape(-1).expand(C) if t.numel() == 1 else t.reshape(-1)
        assert same(outs[0].cpu(), ref), f"call {call}: out vs oracle"
        assert same(st["cmin"].cpu(), vec(ost.cached_min)) and same(st["cmax"].cpu(), vec(ost.cached_max)), f"call {call}"
        assert same(st["scale"].cpu(), vec(ost.scale)) and same(st["rzp"].cpu(), vec(ost.rzp)), f"call {call}: scale"
        assert torch.equal(codes[0].cpu().to(torch.int32), torch.nan_to_num(ref_codes, nan=0.0).to(torch.int32))
        o_out, o_codes, o_st, o_dx = _old_path(xd, gd, bits, layout)
        assert same(outs[0].cpu(), o_out.cpu()) and torch.equal(codes[0], o_codes)
        for k in ("cmin", "cmax", "scale", "zp", "rzp"):
            assert same(st[k].cpu(), o_st[k].cpu()), f"call {call}: {k}"
        dx = F.act_fresh_quant_bwd([gd], [xd], vecs, bits, layout)[0]
        same_act_backward(dx, o_dx, x, st["cmin"], st["cmax"], f"call {call}")
        # and against the oracle's autograd
        xr = x.clone().requires_grad_(True)
        r2, _ = O.act_forward(xr, O.ActState(), bits, "calibration", **kw)
        (ref_dx,) = torch.autograd.grad((r2.float() * g.float()).sum(), [xr])
        ok = torch.isfinite(ref_dx)
        routed = (x == x.amax(0, keepdim=True)) | (x == x.amin(0, keepdim=True))
        z = torch.zeros_like(ref_dx)
        tight = 1e-5 if dtype == torch.float32 else 2e-2
        for sel, tol in ((ok & ~routed, tight), (ok & routed, max(tight, 5e-3))):
            err = rel_err(torch.where(sel, dx.cpu(), z), torch.where(sel, ref_dx, z))
            assert err <= tol, (call, err, tol)


@pytest.mark.parametrize("dt", ["f32", "f16"])
@pytest.mark.parametrize("n", [2, 3, 8])
def test_slabs_of_one_launch_are_independent(dt, n):
    dtype = DT[dt]
    R, C, bits, counts = 260, 512, 4, [64] * 8
    layout = F.ClusterLayout.get(counts, C, DEV)
    xs = [(synth.activation((R, C), 400 + i, dtype) * (1 + i)).to(dtype).to(DEV) for i in range(n)]
    xs[1] = torch.relu(xs[1])
    gs = [synth.grad_like(x.cpu(), 450 + i).to(DEV) for i, x in enumerate(xs)]
    outs, codes, vecs = F.act_fresh_quant_fwd(xs, bits, layout, want_codes=True)
    dxs = F.act_fresh_quant_bwd(gs, xs, vecs, bits, layout)
    for i in range(n):
        o_out, o_codes, o_st, o_dx = _old_path(xs[i], gs[i], bits, layout)
        assert same(outs[i].cpu(), o_out.cpu()) and torch.equal(codes[i], o_codes), f"slab {i}"
        for k, row in (("cmin", 0), ("cmax", 1), ("scale", 2), ("zp", 3), ("rzp", 4)):
            assert same(vecs[i][row].cpu(), o_st[k].cpu()), f"slab {i}: {k}"
        same_act_backward(dxs[i], o_dx, xs[i], vecs[i][0], vecs[i][1], f"slab {i}")


@pytest.mark.parametrize("shape,dt,counts,n", [
    ((2048, 4096), "f32", [128] * 32, 3), ((2048, 4096), "f32", [128] * 32, 2), ((2048, 4096), "f16", [32] * 128, 3),
    ((2048, 11008), "f16", [344] * 32, 2), ((2041, 11008), "f32", [344] * 32, 2), ((1000, 5120), "f16", [160] * 32, 3),
])
def test_llama_slabs_of_one_launch_equal_their_own_launches(shape, dt, counts, n):
    """A launch over several slabs cuts the rows into other chunks than a launch over one slab does: outputs, statistics
    vectors and input gradients stay those of one launch per slab, bit for bit (bench shapes, ragged row counts)."""
    dtype = DT[dt]
    R, C = shape
    layout = F.ClusterLayout.get(counts, C, DEV)
    gen = torch.Generator(device=DEV).manual_seed(77)
    xs = [(torch.randn(R, C, device=DEV, generator=gen) * (1 + i)).to(dtype) for i in range(n)]
    gs = [(torch.randn(R, C, device=DEV, generator=gen) * 1e-3).to(dtype) for _ in range(n)]
    outs, _codes, vecs = F.act_fresh_quant_fwd(xs, 4, layout)
    dxs = F.act_fresh_quant_bwd(gs, xs, vecs, 4, layout)
    for i in range(n):
        o1, _c1, v1 = F.act_fresh_quant_fwd([xs[i]], 4, layout)
        d1 = F.act_fresh_quant_bwd([gs[i]], [xs[i]], v1, 4, layout)
        assert same(outs[i].cpu(), o1[0].cpu()), f"slab {i}: out"
        assert same(vecs[i].cpu(), v1[0].cpu()), f"slab {i}: statistics"
        assert same(dxs[i].cpu(), d1[0].cpu()), f"slab {i}: dx"


LLAMA = [  # Llama-2 7B / 13B / 70B activation sites of one 2048-token sample
    ((2048, 4096), "f32", [128] * 32), ((2048, 4096), "f16", [32] * 128), ((2048, 11008), "f16", [344] * 32),
    ((2048, 5120), "f32", [160] * 32), ((2048, 13824), "f16", [432] * 32),
    ((2048, 8192), "f32", [256] * 32), ((2048, 8192), "f16", [32] * 256), ((2048, 28672), "f16", [896] * 32),
    ((2048, 28672), "f32", None),
]


@pytest.mark.parametrize("shape,dt,counts", LLAMA)
@pytest.mark.parametrize("bits", [4, 8])
def test_llama_shapes_fresh_equals_four_kernel_path(shape, dt, counts, bits):
    dtype = DT[dt]
    R, C = shape
    layout = F.ClusterLayout.get(counts, C, DEV)
    x = synth.activation(shape, 500, dtype).to(DEV)
    g = (torch.randn(shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5)) * 1e-3).to(dtype)
    outs, _, vecs = F.act_fresh_quant_fwd([x], bits, layout)
    dx = F.act_fresh_quant_bwd([g], [x], vecs, bits, layout)[0]
    o_out, _, o_st, o_dx = _old_path(x, g, bits, layout)
    assert torch.equal(outs[0], o_out)
    for k, row in (("cmin", 0), ("cmax", 1), ("scale", 2), ("zp", 3), ("rzp", 4)):
        assert torch.equal(vecs[0][row], o_st[k]), k
    same_act_backward(dx, o_dx, x, vecs[0][0], vecs[0][1])
    # size-independent properties: quantizing the output again with the same parameters changes nothing, at most
    # 2^bits distinct levels per channel, and the codes reproduce the output
    again, codes = F.colquant_fwd(outs[0], vecs[0][2], vecs[0][4], bits, want_codes=True)
    assert torch.equal(again, outs[0])
    assert int(codes.max()) <= (1 << bits) - 1
    deq = ((codes.float() - vecs[0][4].float()) * vecs[0][2].float()).to(dtype)
    assert torch.equal(deq, outs[0])


def test_oracle_at_llama_70b_widths():
    """The oracle itself at the Llama-2-70B widths C = 8192 and C = 28672, up to the full 2048-token slab."""
    for R, C, counts, dt in ((256, 8192, [256] * 32, torch.float32), (256, 28672, [896] * 32, torch.float16),
                             (256, 8192, None, torch.float16), (2048, 8192, [256] * 32, torch.float16),
                             (2048, 28672, [896] * 32, torch.float32)):
        x = synth.activation((R, C), 77, dt)
        layout = F.ClusterLayout.get(counts, C, DEV)
        outs, codes, vecs = F.act_fresh_quant_fwd([x.to(DEV)], 8, layout, want_codes=True)
        kw = dict(method="per_cluster" if counts else "per_tensor", metric="ema_minmax", cluster_dim=1 if counts else None,
                  cluster_counts=counts)
        ref, ref_codes = O.act_forward(x, O.ActState(), 8, "calibration", **kw)
        assert same(outs[0].cpu(), ref)
        assert torch.equal(codes[0].cpu().to(torch.int32), ref_codes.to(torch.int32))


@pytest.mark.parametrize("dt", ["f32", "f16"])
@pytest.mark.parametrize("shape", [(64, 64), (96, 2048)])
def test_tied_extremes_share_the_scale_gradient_like_autograd(dt, shape):
    """Three elements tie at the minimum (two of them in one channel) and three at the maximum.  Per-tensor statistics
    are one reduction over all elements (torch.min / torch.max, quantizer.py:222-223): autograd gives every tied element
    a third of the scale gradient.  Statistics of an explicit cluster are found per channel first (quantizer.py:190-210):
    the two tied channels get half each, and the channel with two tied rows splits its half again."""
    dtype = DT[dt]
    R, C = shape
    x = synth.activation(shape, 900, dtype)
    lo, hi = x.min() - 1, x.max() + 1
    x[0, 44] = x[0, 61] = x[1, 44] = lo
    x[5, 3] = x[9, 3] = x[2, 17] = hi
    g = synth.grad_like(x, 901)
    xd, gd = x.to(DEV), g.to(DEV)
    tied = (x == lo) | (x == hi)
    for counts, kw in ((None, dict(method="per_tensor", metric="minmax")),
                       ([C], dict(method="per_cluster", metric="minmax", cluster_dim=1, cluster_counts=[C]))):
        layout = F.ClusterLayout.get(counts, C, DEV)
        assert layout.flat == (counts is None)
        xr = x.clone().requires_grad_(True)
        ref, _ = O.act_forward(xr, O.ActState(), 4, "calibration", **kw)
        (ref_dx,) = torch.autograd.grad((ref.float() * g.float()).sum(), [xr])
        outs, _, vecs = F.act_fresh_quant_fwd([xd], 4, layout)
        assert same(outs[0].cpu(), ref.detach())
        dx = F.act_fresh_quant_bwd([gd], [xd], vecs, 4, layout)[0].cpu()
        _, _, o_st, o_dx = _old_path(xd, gd, 4, layout)
        same_act_backward(dx, o_dx, x, vecs[0][0], vecs[0][1], str(counts))
        got, want = dx[tied].double(), ref_dx[tied].double()
        tol = 1e-5 if dtype == torch.float32 else 2e-2
        assert ((got - want).abs().max() / want.abs().max()).item() <= tol, (counts, got, want)
        if counts is None:  # the two ways of splitting really differ on this input
            layout_h = F.ClusterLayout.get([C], C, DEV)
            dx_h = F.act_fresh_quant_bwd([gd], [xd], vecs, 4, layout_h)[0].cpu()
            assert ((dx_h[tied].double() - want).abs().max() / want.abs().max()).item() > 0.05


def test_autograd_function_uses_the_fresh_path_and_the_batch_function_agrees():
    R, C, bits = 192, 512, 4
    layout = F.ClusterLayout.get([128] * 4, C, DEV)
    xs = [synth.activation((2, R // 2, C), 600 + i, torch.float16).to(DEV).requires_grad_(True) for i in range(3)]
    gs = [synth.grad_like(x.detach().cpu(), 650 + i).to(DEV) for i, x in enumerate(xs)]
    singles, holders = [], []
    for x in xs:
        h = {}
        singles.append(F.ActCalibFQ.apply(x, bits, "ema_minmax", layout, None, None, h, None))
        holders.append(h)
        assert "vec" in h and h["scale"].shape == (C,)
    d_single = torch.autograd.grad(singles, xs, gs)
    hb = [{} for _ in xs]
    outs = F.ActCalibFQBatch.apply(bits, layout, hb, *xs)
    d_batch = torch.autograd.grad(outs, xs, gs)
    for i in range(3):
        assert torch.equal(outs[i], singles[i]) and torch.equal(d_batch[i], d_single[i])
        assert torch.equal(hb[i]["scale"], holders[i]["scale"]) and torch.equal(hb[i]["cmax"], holders[i]["cmax"])


def test_bad_arguments():
    lib = L.lib()
    assert int(lib.miq_act_fresh_workspace(L.i32(0), L.i64(8), L.i32(1), L.i64(8))) == 0
    x = torch.zeros((4, 16), device=DEV)
    layout = F.ClusterLayout.get([8, 8], 16, DEV)
    with pytest.raises(L.MiqError):
        F.act_fresh_quant_fwd([x] * 9, 4, layout)
    with pytest.raises(L.MiqError):
        F.act_fresh_quant_fwd([x, torch.zeros((4, 16), device=DEV, dtype=torch.float16)], 4, layout)


@pytest.mark.parametrize("dt", ["f32", "f16"])
@pytest.mark.parametrize("counts", [[128] * 4, None])
def test_infinite_nan_and_huge_activations_match_the_oracle(dt, counts):
    """Channels that hold an infinity, a NaN, the largest finite value or only zeros: the clamped scale keeps the
    corrected quotient in use, which must still give the reference's outputs, codes and statistics."""
    dtype = DT[dt]
    R, C = 96, 512
    layout = F.ClusterLayout.get(counts, C, DEV)
    kw = dict(method="per_cluster" if counts else "per_tensor", metric="minmax", cluster_dim=1 if counts else None,
              cluster_counts=counts)
    for special in (float("inf"), -float("inf"), float("nan"), torch.finfo(dtype).max, 0.0):
        x = synth.activation((R, C), 77, dtype)
        x[:, 130:140] = 0.0
        if special == 0.0:
            x[:, 256:384] = 0.0      # one whole cluster of zeros
        else:
            x[5, 3] = special        # first cluster
            x[70, 300] = special     # third cluster
        xd = x.to(DEV)
        outs, codes, vecs = F.act_fresh_quant_fwd([xd], 4, layout, want_codes=True)
        st = F.stats_of(vecs[0])
        ref, ref_codes = O.act_forward(x, (ost := O.ActState()), 4, "calibration", **kw)
        vec = lambda t: t.reshape(-1).expand(C) if t.numel() == 1 else t.reshape(-1)
        assert same(st["scale"].cpu(), vec(ost.scale)) and same(st["rzp"].cpu(), vec(ost.rzp)), special
        assert same(outs[0].cpu(), ref), special
        assert torch.equal(codes[0].cpu().to(torch.int32), torch.nan_to_num(ref_codes, nan=0.0).to(torch.int32)), special
        o_out, o_codes, o_st = F.act_calib_quant_fwd(xd, 4, "minmax", layout, want_codes=True)
        assert same(o_out.cpu(), ref) and torch.equal(o_codes, codes[0]), special


@pytest.mark.parametrize("dt", ["f32", "f16"])
def test_signed_zero_extremes_give_the_same_bits_on_both_paths(dt):
    """A cluster whose minimum (or maximum) is zero, present as -0 and +0: float comparisons keep whichever zero came
    first, integer keys order -0 below +0; both paths report +0, so statistics, zero points and outputs agree in every
    bit, sign of zero included."""
    dtype = DT[dt]
    R, C, bits, counts = 96, 256, 4, [64] * 4
    layout = F.ClusterLayout.get(counts, C, DEV)
    x = torch.relu(synth.activation((R, C), 910, dtype)).to(dtype)
    x[::3] = x[::3] * -0.0 + torch.where(x[::3] > 0, x[::3], torch.zeros_like(x[::3]))  # -0 where relu cut, values kept
    x[1::3, :64] = -torch.relu(-synth.activation((R, C), 911, dtype))[1::3, :64].to(dtype)  # cluster 0: maximum zero
    x[0, 5] = -0.0
    x[2, 5] = 0.0
    xd = x.to(DEV)
    outs, _codes, vecs = F.act_fresh_quant_fwd([xd], bits, layout)
    o_out, _c, o_st = F.act_calib_quant_fwd(xd, bits, "minmax", layout)
    bits_of = lambda t: t.contiguous().view(torch.int32 if t.dtype == torch.float32 else torch.int16)
    assert torch.equal(bits_of(outs[0]), bits_of(o_out))
    for k, row in (("cmin", 0), ("cmax", 1), ("scale", 2), ("zp", 3), ("rzp", 4)):
        assert torch.equal(bits_of(vecs[0][row]), bits_of(o_st[k])), k
    for row in (0, 1):  # a zero extreme is +0
        assert not (torch.signbit(vecs[0][row]) & (vecs[0][row] == 0)).any()
