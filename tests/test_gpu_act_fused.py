"""One-call forward of the calibrating activation quantizer (miq_act_calib_quant_fwd) against the
oracle (bit-identical forward incl. cached statistics and integer codes; 1e-5 / 2e-2 backward) and
against the three-call chain, plus the phased backward (phase 1 + phase 2, the data-parallel
form) against the single-call backward (phase 0), which must agree bit for bit.

Shapes exercise ragged row chunks, partial column tiles, clusters of 1 channel and of >64 channels,
channels outside every cluster, and cached statistics (running min/max and EMA).  The fp16 EMA case
[2048, 4096] holds a value whose product with 0.99 lies exactly on an fp16 midpoint after the fp32
rounding: it only matches when the kernel rounds twice (fp32, then fp16) like PyTorch.
"""
import pytest
import torch

from helpers import DT, rel_err, same
from miq import _lib as L
from miq import functional as F
from miq import synth
from oracle import miq_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = [
    # shape, dtype, bits, metric, cluster counts (None = one cluster over all channels)
    ((2048, 4096), "f32", 4, "ema_minmax", [128] * 32),
    ((2048, 4096), "f16", 4, "ema_minmax", [128] * 32),
    ((512, 11008), "f16", 8, "minmax", [344] * 32),
    ((333, 1000), "f32", 4, "minmax", [50] * 10 + [20] * 10 + [7, 93]),  # 100 channels outside every cluster
    ((77, 520), "f16", 6, "ema_minmax", [1] * 20 + [400]),  # single-channel clusters, 100 left over
    ((333, 1000), "f32", 4, "minmax", [100, 300, 7, 293, 200]),
    ((5, 64), "f32", 3, "minmax", [64]),
    ((1, 8), "f16", 4, "minmax", [3, 5]),
    ((4100, 256), "f32", 4, "minmax", None),
    ((64, 100), "f16", 5, "ema_minmax", [5] * 20),  # C not a multiple of the 8-element pack
]


def _phased_bwd(g, x, st, bits, layout):
    """miq_colquant_bwd as phase 1 (stream + fold) then phase 2 (shares + correction)."""
    R, C = x.shape
    nchunk = F.colstat_chunks(R, C, x.dtype)
    ws = torch.empty((3 * nchunk + 5) * C, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    for phase in (1, 2):
        L.call("miq_colquant_bwd", L.ptr(g), L.ptr(x), L.ptr(st["scale"]), L.ptr(st["rzp"]), L.ptr(st["chan_min"]),
               L.ptr(st["chan_max"]), L.ptr(st["cmin"]), L.ptr(st["cmax"]), L.ptr(layout.starts), L.i32(layout.n_clusters),
               L.i64(layout.covered), L.ptr(ws), L.i32(nchunk), L.ptr(dx), L.i64(R), L.i64(C), L.i32(bits), L.i32(phase),
               L.i32(int(layout.flat)), L.i32(L.dtype_code(x)), L.stream_of(x))
    return dx


@pytest.mark.parametrize("shape,dt,bits,metric,counts", CASES)
def test_one_call_forward_and_phased_backward(shape, dt, bits, metric, counts):
    dtype = DT[dt]
    R, C = shape
    layout = F.ClusterLayout.get(counts, C, DEV)
    ostate = O.ActState()
    kw = dict(method="per_cluster" if counts else "per_tensor", metric=metric, cluster_dim=1 if counts else None,
              cluster_counts=counts)
    cmin = cmax = None
    for call in range(3):
        x = synth.activation(shape, 300 + call, dtype)
        if call == 2:
            x = torch.relu(x)  # many ties at the minimum
        xd = x.to(DEV)
        out, codes, st = F.act_calib_quant_fwd(xd, bits, metric, layout, cmin, cmax, want_codes=True)
        ref, ref_codes = O.act_forward(x, ostate, bits, "calibration", **kw)
        assert same(out.cpu(), ref), f"call {call}: out"
        vec = lambda t: t.reshape(-1).expand(C) if t.numel() == 1 else t.reshape(-1)
        assert same(st["cmin"].cpu(), vec(ostate.cached_min)) and same(st["cmax"].cpu(), vec(ostate.cached_max)), f"call {call}: cache"
        assert same(st["scale"].cpu(), vec(ostate.scale)) and same(st["rzp"].cpu(), vec(ostate.rzp)), f"call {call}: scale"
        assert torch.equal(codes.cpu().to(torch.int32), torch.nan_to_num(ref_codes, nan=0.0).to(torch.int32)), f"call {call}"
        # the three-call chain gives the same bits
        st2 = F.act_calibrate(xd, bits, metric, layout, cmin, cmax)
        out2, _ = F.colquant_fwd(xd, st2["scale"], st2["rzp"], bits)
        assert same(out.cpu(), out2.cpu())
        for k in ("cmin", "cmax", "scale", "zp", "rzp", "chan_min", "chan_max"):
            assert same(st[k].cpu(), st2[k].cpu()), f"call {call}: {k}"
        cmin, cmax = st["cmin"], st["cmax"]
    # backward with fresh statistics: phase 0 == phases 1 + 2, and ~= oracle autograd
    for seed, relu in ((310, False), (311, True)):
        x = synth.activation(shape, seed, dtype)
        x = torch.relu(x) if relu else x
        g = synth.grad_like(x, seed + 50)
        xd, gd = x.to(DEV), g.to(DEV)
        out, _, st = F.act_calib_quant_fwd(xd, bits, metric, layout)
        dx = F.colquant_bwd(gd, xd, st["scale"], st["rzp"], bits,
                            route=(st["chan_min"], st["chan_max"], st["cmin"], st["cmax"], layout))
        assert same(dx.cpu(), _phased_bwd(gd, xd, st, bits, layout).cpu()), "single-call vs phased backward"
        xr = x.clone().requires_grad_(True)
        ref, _ = O.act_forward(xr, O.ActState(), bits, "calibration", **kw)
        (ref_dx,) = torch.autograd.grad((ref.float() * g.float()).sum(), [xr])
        ok = torch.isfinite(ref_dx)
        # elements that attain their channel's min / max also carry the scale gradient, a sum over
        # the whole cluster slab whose fp32 rounding depends on the summation order: looser bound there
        routed = (x == x.amax(0, keepdim=True)) | (x == x.amin(0, keepdim=True))
        z = torch.zeros_like(ref_dx)
        tight = 1e-5 if dtype == torch.float32 else 2e-2
        for sel, tol in ((ok & ~routed, tight), (ok & routed, max(tight, 5e-3))):
            err = rel_err(torch.where(sel, dx.cpu(), z), torch.where(sel, ref_dx, z))
            assert err <= tol, (err, tol)
        if dtype == torch.float32:  # fp16: the reference's (g*s)/s can underflow to 0 where the kernel keeps g
            assert torch.equal((dx.cpu() == 0) | ~ok, (ref_dx == 0) | ~ok)


def test_workspace_size_matches_chunks():
    for (R, C, dt) in ((2048, 4096, torch.float32), (7, 24, torch.float16)):
        n = int(L.lib().miq_act_workspace(L.i64(R), L.i64(C), L.i32(L._DT[dt])))
        assert n == 2 * (F.colstat_chunks(R, C, dt) + 1) * C


@pytest.mark.parametrize("dt", ["f32", "f16"])
@pytest.mark.parametrize("bad", ["inf", "nan"])
def test_backward_with_non_finite_activations(dt, bad):
    """+-inf / NaN activations: the statistics, scales and the backward follow PyTorch's propagation rules
    (inf saturates the clamped scale and turns its own level into NaN through round_ste; NaN poisons the
    statistics of its cluster).  The positions of NaN in dx and the finite entries must match autograd."""
    dtype = DT[dt]
    R, C, bits = 96, 256, 4
    counts = [64] * 4
    layout = F.ClusterLayout.get(counts, C, DEV)
    x = synth.activation((R, C), 77, dtype)
    if bad == "inf":
        x[3, 5] = float("inf")
        x[40, 200] = float("-inf")
    else:
        x[3, 5] = float("nan")
    g = synth.grad_like(x, 78)
    xd, gd = x.to(DEV), g.to(DEV)
    out, _, st = F.act_calib_quant_fwd(xd, bits, "minmax", layout)
    xr = x.clone().requires_grad_(True)
    ref, _ = O.act_forward(xr, O.ActState(), bits, "calibration", method="per_cluster", metric="minmax", cluster_dim=1,
                           cluster_counts=counts)
    assert same(out.cpu(), ref.detach())
    dx = F.colquant_bwd(gd, xd, st["scale"], st["rzp"], bits,
                        route=(st["chan_min"], st["chan_max"], st["cmin"], st["cmax"], layout)).cpu()
    (ref_dx,) = torch.autograd.grad((ref.float() * g.float()).sum(), [xr])
    assert torch.equal(torch.isnan(dx), torch.isnan(ref_dx))
    ok = torch.isfinite(ref_dx)
    z = torch.zeros_like(ref_dx)
    assert rel_err(torch.where(ok, dx, z), torch.where(ok, ref_dx, z)) <= (5e-3 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("dt", DT)
def test_narrow_per_tensor_slabs_are_folded_without_changing_the_result(dt):
    """[.., head_dim] operands with per-tensor statistics run through the column kernels as wide rows: the
    quantized values are the same bit for bit, the gradient equal up to the summation order of the scale term."""
    dt = DT[dt]
    g = torch.Generator(device=DEV).manual_seed(9)
    x = (torch.randn(2, 4, 96, 128, generator=g, device=DEV) * 2).to(dt)
    go = torch.randn(x.shape, generator=g, device=DEV).to(dt)
    layout = F.ClusterLayout.get(None, 128, DEV)
    assert F._tensor_fold(2 * 4 * 96, 128, layout) == 32
    assert F._tensor_fold(2 * 4 * 96, 128, F.ClusterLayout.get([64, 64], 128, DEV)) == 1
    assert F._tensor_fold(7, 128, layout) == 1 and F._tensor_fold(6, 128, layout) == 2
    x2 = x.reshape(-1, 128)
    ref_out, _, ref_st = F.act_calib_quant_fwd(x2, 4, "minmax", layout)
    ref_dx = F.colquant_bwd(go.reshape(-1, 128), x2, ref_st["scale"], ref_st["rzp"], 4,
                            route=(ref_st["chan_min"], ref_st["chan_max"], ref_st["cmin"], ref_st["cmax"], layout))
    leaf = x.clone().requires_grad_(True)
    holder = {}
    out = F.ActCalibFQ.apply(leaf, 4, "minmax", layout, None, None, holder, None)
    assert torch.equal(out.reshape(-1, 128), ref_out)
    for name in ("scale", "zp", "rzp", "cmin", "cmax"):
        assert holder[name].shape == (128,) and torch.equal(holder[name], ref_st[name])
    (dx,) = torch.autograd.grad(out, leaf, go)
    assert rel_err(dx.reshape(-1, 128), ref_dx) <= (2e-3 if dt == torch.float16 else 1e-5)
