"""Norm kernels that publish the statistics of their own output (miq_rmsnorm_fwd_stats, miq_gather_layernorm_fwd_stats)
followed by the quantize half of the fresh-statistics activation quantizer (miq_act_quant_from_table), and the HIP
backward of gather + LayerNorm (miq_gather_layernorm_bwd).

Bars: the fused pair returns bit for bit what the norm followed by miq_act_fresh_quant_fwd returns on the norm's own
output (quantized tensor, merged minima / maxima, scale, zero points), for per-tensor statistics, equal clusters,
unequal clusters, clusters that are no multiple of the 4-channel pack and channels outside every cluster, rows that do
not fill the last block, teams narrower than a wave; its backward equals the backward of the two separate functions bit
for bit (same kernels on the same tensors).  The LayerNorm backward agrees with autograd through index_select +
layer_norm in float64 to 2e-5 of the gradient's largest magnitude (fp32) / 2e-3 (fp16 activations).
"""
import pytest
import torch

from helpers import DT
from miq import functional as F
from miq import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [
    # rows, channels, dtype, cluster counts
    (300, 512, "f32", None),
    (300, 512, "f32", [128] * 4),
    (77, 64, "f16", [16] * 4),               # four rows per wave
    (77, 64, "f32", None),
    (130, 4096, "f32", [128] * 32),
    (130, 4096, "f16", [32] * 128),
    (35, 4096, "f16", None),
    (50, 11008, "f16", [344] * 32),
    (64, 1000, "f32", [50] * 10 + [20] * 10 + [7, 93]),   # unequal clusters, 100 channels outside every cluster
    (64, 1000, "f32", [10] * 100),                        # equal clusters, no multiple of the pack
    (3, 256, "f16", [1] * 256),
]


def _vectors(C, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    w = (1 + 0.2 * torch.randn(C, generator=g)).to(DEV)
    b = (0.1 * torch.randn(C, generator=g)).to(DEV)
    return w, b


@pytest.mark.parametrize("R,C,dt,counts", SHAPES)
@pytest.mark.parametrize("with_bias", [False, True])
def test_rmsnorm_publishes_the_statistics_of_its_output(R, C, dt, counts, with_bias):
    dtype = DT[dt]
    layout = F.ClusterLayout.get(counts, C, DEV)
    w, b = _vectors(C, dtype, 5)
    b = b if with_bias else None
    for call, gain in enumerate((1.0, 3.0, 0.25)):   # the table is reused: stale statistics must not survive
        x = (synth.activation((R, C), 600 + call, dtype) * gain).to(dtype).to(DEV)
        xa = x.clone().requires_grad_(True)
        wa, ba = w.clone().requires_grad_(True), (None if b is None else b.clone().requires_grad_(True))
        y = F.RMSNorm.apply(xa, wa, ba, 1e-6)
        outs, _, vecs = F.act_fresh_quant_fwd([y.detach().reshape(R, C)], 4, layout)
        xb = x.clone().requires_grad_(True)
        wb, bb = w.clone().requires_grad_(True), (None if b is None else b.clone().requires_grad_(True))
        holder = {}
        q = F.RMSNormActQuant.apply(xb, wb, bb, 1e-6, 4, layout, holder)
        assert torch.equal(q, outs[0]), f"call {call}"
        assert torch.equal(holder["vec"], vecs[0]), f"call {call}: statistics differ from the statistics pass on the norm output"
        # backward against the two separate functions
        g = synth.grad_like(x.cpu(), 650 + call).to(DEV)
        F.TENSOR_FOLD, keep = False, F.TENSOR_FOLD   # (narrow per-tensor slabs are otherwise reshaped: other summation order)
        try:
            qa = F.ActCalibFQ.apply(y, 4, "minmax", layout, None, None, {})
            ga = torch.autograd.grad((qa.float() * g.float()).sum(), [xa, wa] + ([ba] if with_bias else []))
        finally:
            F.TENSOR_FOLD = keep
        gb = torch.autograd.grad((q.float() * g.float()).sum(), [xb, wb] + ([bb] if with_bias else []))
        for u, v in zip(ga, gb):
            assert torch.equal(u, v), f"call {call}"


def _ln_ref(x, idx, w, b, eps, out_dtype):
    xg = torch.index_select(x, -1, idx.long())
    return torch.nn.functional.layer_norm(xg.float(), (x.shape[-1],), w, b, eps).to(out_dtype)


@pytest.mark.parametrize("R,C,dt,counts", SHAPES[:9])
@pytest.mark.parametrize("amp_out", [False, True])
def test_gather_layernorm_publishes_the_statistics_of_its_output(R, C, dt, counts, amp_out):
    dtype = DT[dt]
    if C > 8192:
        pytest.skip("the fused gather + LayerNorm kernel serves hidden sizes only")
    out_dtype = torch.float32 if amp_out else dtype
    if C % (8 if out_dtype == torch.float16 else 4):
        pytest.skip("the quantize kernel needs whole packs")
    layout = F.ClusterLayout.get(counts, C, DEV)
    w, b = _vectors(C, dtype, 7)
    w, b = w.to(dtype), b.to(dtype)
    idx = synth.permutation(C, 9).to(torch.int32).to(DEV)
    inv = torch.argsort(idx.long()).to(torch.int32)
    for call, gain in enumerate((1.0, 2.0, 0.5)):
        x = (synth.activation((R, C), 700 + call, dtype) * gain).to(dtype).to(DEV)
        xa, wa, ba = (t.clone().requires_grad_(True) for t in (x, w, b))
        y = F.GatherLayerNorm.apply(xa, idx, inv, wa, ba, 1e-5, out_dtype)
        outs, _, vecs = F.act_fresh_quant_fwd([y.detach().reshape(R, C)], 4, layout)
        xb, wb, bb = (t.clone().requires_grad_(True) for t in (x, w, b))
        holder = {}
        q = F.GatherLayerNormActQuant.apply(xb, idx, wb, bb, 1e-5, out_dtype, 4, layout, holder)
        assert q.dtype == out_dtype and torch.equal(q, outs[0]), f"call {call}"
        assert torch.equal(holder["vec"], vecs[0]), f"call {call}"
        g = synth.grad_like(y.detach().cpu(), 750 + call).to(DEV)
        F.TENSOR_FOLD, keep = False, F.TENSOR_FOLD
        try:
            qa = F.ActCalibFQ.apply(y, 4, "minmax", layout, None, None, {})
            ga = torch.autograd.grad((qa.float() * g.float()).sum(), [xa, wa, ba])
        finally:
            F.TENSOR_FOLD = keep
        gb = torch.autograd.grad((q.float() * g.float()).sum(), [xb, wb, bb])
        for u, v in zip(ga, gb):
            assert torch.equal(u, v), f"call {call}"


@pytest.mark.parametrize("R,C,dt", [(300, 512, "f32"), (77, 64, "f16"), (130, 4096, "f32"), (130, 4096, "f16"), (33, 768, "f32"),
                                    (2048, 4096, "f16")])
@pytest.mark.parametrize("with_idx", [True, False])
@pytest.mark.parametrize("amp_out", [False, True])
def test_gather_layernorm_backward_matches_autograd(R, C, dt, with_idx, amp_out):
    dtype = DT[dt]
    out_dtype = torch.float32 if amp_out else dtype
    w, b = _vectors(C, dtype, 11)
    w, b = w.to(dtype), b.to(dtype)
    idx = synth.permutation(C, 13).to(torch.int32).to(DEV) if with_idx else None
    inv = torch.argsort(idx.long()).to(torch.int32) if with_idx else None
    x = synth.activation((R, C), 800, dtype).to(DEV)
    g = synth.grad_like(x.cpu().float(), 801).to(DEV).to(out_dtype)
    xa, wa, ba = (t.clone().requires_grad_(True) for t in (x, w, b))
    y = F.GatherLayerNorm.apply(xa, idx, inv, wa, ba, 1e-5, out_dtype)
    got = torch.autograd.grad((y.float() * g.float()).sum(), [xa, wa, ba])
    # float64 autograd through index_select + layer_norm
    xr, wr, br = (t.double().clone().requires_grad_(True) for t in (x, w, b))
    xg = torch.index_select(xr, -1, idx.long()) if with_idx else xr
    yr = torch.nn.functional.layer_norm(xg, (C,), wr, br, 1e-5)
    ref = torch.autograd.grad((yr * g.double()).sum(), [xr, wr, br])
    tol = 2e-5 if dtype == torch.float32 else 2e-3
    for name, u, v in zip(("dx", "d_weight", "d_bias"), got, ref):
        assert u.dtype == dtype
        err = (u.double() - v).abs().max().item() / max(v.abs().max().item(), 1e-30)
        assert err <= tol, (name, err)
    # only some of the gradients wanted
    xa2 = x.clone().requires_grad_(True)
    y2 = F.GatherLayerNorm.apply(xa2, idx, inv, w, b, 1e-5, out_dtype)
    (dx_only,) = torch.autograd.grad((y2.float() * g.float()).sum(), [xa2])
    assert torch.equal(dx_only, got[0])


def test_bad_arguments_are_refused():
    from miq import _lib as L
    x = torch.randn(8, 64, device=DEV)
    w = torch.ones(64, device=DEV)
    y = torch.empty_like(x)
    ws = torch.zeros(1 << 16, dtype=torch.uint8, device=DEV)
    with pytest.raises(L.MiqError):   # unequal clusters without the slot map
        L.call("miq_rmsnorm_fwd_stats", L.ptr(x), L.ptr(w), None, L.ptr(y), None, L.i64(8), L.i64(64), L.f32(1e-6), L.i32(L.F32),
               L.i64(0), None, L.i32(3), L.i64(60), L.ptr(ws), L.stream_of(x))
    with pytest.raises(L.MiqError):   # misaligned workspace
        L.call("miq_rmsnorm_fwd_stats", L.ptr(x), L.ptr(w), None, L.ptr(y), None, L.i64(8), L.i64(64), L.f32(1e-6), L.i32(L.F32),
               L.i64(64), None, L.i32(1), L.i64(64), L.ptr(ws[8:]), L.stream_of(x))
    with pytest.raises(L.MiqError):   # cluster_size * n_clusters != covered
        L.call("miq_act_quant_from_table", L.ptr(x), L.i64(16), None, L.i32(3), L.i64(64), L.i32(4), L.ptr(ws), L.ptr(y), None,
               L.ptr(torch.empty(5, 64, device=DEV)), L.i64(8), L.i64(64), L.i32(L.F32), L.stream_of(x))
