"""GPU parity of the static activation quantizer (statistics, cluster merge, scale, quantize,
backward) and of the reorder gather, through the drop-in module and the C ABI.

Everything on this path is IEEE add/sub/mul/div/round/compare, so the forward results (outputs,
scales, zero points, cached statistics, integer codes) must be BIT-IDENTICAL to the reference's
golden vectors.  Gradients: norm-wise relative error <= 1e-5 (fp32) / 2e-2 (fp16, the reference
sums in fp16).
"""
import numpy as np
import pytest
import torch

from helpers import DT, golden, rel_err, same
from miq import functional as F
from miq import synth
from miq.quantizer import UniformAffineQuantizer
from oracle import miq_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make_q(m):
    q = UniformAffineQuantizer(n_bits=m["n_bits"], symmetric=False, per_channel_axes=[], metric=m["metric"],
                               dynamic=False, dynamic_method=m["method"])
    q.cluster_dim = None if m["cluster_dim"] < 0 else m["cluster_dim"]
    q.cluster_counts = np.array(m["counts"]) if m["counts"] else None
    return q


def masked_rel(a, ref):
    ok = torch.isfinite(ref)
    z = torch.zeros_like(ref)
    return rel_err(torch.where(ok, a.cpu(), z), torch.where(ok, ref, z))


@pytest.mark.parametrize("name", sorted(golden("act").keys()))
def test_act_golden(name):
    d = golden("act")[name]
    m = d["meta"]
    dtype, shape, seed = DT[m["dtype"]], tuple(m["shape"]), m["seed"]
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    q = make_q(m)
    for c in range(m["calls"]):
        x = synth.activation(shape, seed + c, dtype).to(DEV)
        if c == 0:
            xg = x.clone().requires_grad_(True)
            out = q(xg)
            w = synth.grad_like(x.cpu(), seed + 100).to(DEV)
            (dx,) = torch.autograd.grad((out.float() * w.float()).sum(), [xg])
            assert masked_rel(dx, d["dx0"]) <= tol, masked_rel(dx, d["dx0"])
            # same zero pattern (mask + routing touch exactly the same elements)
            assert torch.equal(dx.cpu() == 0, d["dx0"] == 0)
            q.free()
        with torch.no_grad():
            out = q(x)
        assert same(out.cpu(), d[f"out{c}"]), f"call {c}: output"
        assert same(q.scale.cpu(), d[f"scale{c}"]), f"call {c}: scale"
        assert same(q.zero_point.cpu(), d[f"zp{c}"]), f"call {c}: zero point"
        assert same(q.round_zero_point.cpu(), d[f"rzp{c}"]), f"call {c}: rounded zero point"
        assert same(q.cached_xmin.cpu(), d[f"cmin{c}"]) and same(q.cached_xmax.cpu(), d[f"cmax{c}"]), f"call {c}: cache"
    q.set_eval_mode()
    x = synth.activation(shape, seed + 77, dtype).to(DEV)
    xg = x.clone().requires_grad_(True)
    out = q(xg)
    w = synth.grad_like(x.cpu(), seed + 101).to(DEV)
    (dx,) = torch.autograd.grad((out.float() * w.float()).sum(), [xg])
    assert same(out.detach().cpu(), d["out_eval"])
    assert masked_rel(dx, d["dx_eval"]) <= tol
    assert torch.equal(dx.cpu() == 0, d["dx_eval"] == 0)


@pytest.mark.parametrize("shape,dt,bits,metric,counts", [
    ((1, 2048, 4096), "f32", 4, "ema_minmax", [128] * 32),
    ((1, 2048, 4096), "f16", 4, "ema_minmax", [32] * 128),
    ((2048, 11008), "f32", 4, "ema_minmax", [344] * 32),
    ((2048, 11008), "f16", 8, "minmax", [344] * 32),
    ((2, 2048, 4096), "f32", 8, "minmax", None),
    ((3, 1000, 5120), "f16", 4, "minmax", [160] * 32),
])
def test_act_llama_shapes_vs_oracle(shape, dt, bits, metric, counts):
    """Llama-2 activation shapes: two calibration calls then eval, bit-identical to the oracle."""
    dtype = DT[dt]
    m = {"n_bits": bits, "metric": metric, "method": "per_cluster" if counts else "per_tensor",
         "cluster_dim": len(shape) - 1 if counts else -1, "counts": counts or []}
    q = make_q(m)
    st = O.ActState()
    kw = dict(method=m["method"], metric=metric, cluster_dim=None if not counts else len(shape) - 1, cluster_counts=counts)
    for c in range(2):
        x = synth.activation(shape, 90 + c, dtype)
        with torch.no_grad():
            out = q(x.to(DEV))
            ref, codes = O.act_forward(x, st, bits, "calibration", **kw)
        assert same(out.cpu(), ref), f"call {c}"
        assert same(q.scale.cpu(), st.scale) and same(q.round_zero_point.cpu(), st.rzp)
        s, z = q._vec[0], q._vec[1]
        _, gcodes = F.colquant_fwd(x.to(DEV).reshape(-1, shape[-1]), s, z, bits, want_codes=True)
        assert torch.equal(gcodes.cpu().to(torch.int32).reshape(shape), torch.nan_to_num(codes, nan=0.0).to(torch.int32))
    q.set_eval_mode()
    x = synth.activation(shape, 99, dtype)
    with torch.no_grad():
        out = q(x.to(DEV))
        ref, _ = O.act_forward(x, st, bits, "eval")
    assert same(out.cpu(), ref)


def test_act_backward_llama_shape_vs_oracle():
    shape, bits, counts = (1, 2048, 4096), 4, [128] * 32
    x = synth.activation(shape, 31)
    w = synth.grad_like(x, 32)
    q = make_q({"n_bits": bits, "metric": "ema_minmax", "method": "per_cluster", "cluster_dim": 2, "counts": counts})
    xg = x.to(DEV).requires_grad_(True)
    (dx,) = torch.autograd.grad((q(xg) * w.to(DEV)).sum(), [xg])
    xr = x.clone().requires_grad_(True)
    out, _ = O.act_forward(xr, O.ActState(), bits, "calibration", method="per_cluster", metric="ema_minmax",
                           cluster_dim=2, cluster_counts=counts)
    (ref,) = torch.autograd.grad((out * w).sum(), [xr])
    assert rel_err(dx, ref) <= 1e-5
    assert torch.equal(dx.cpu() == 0, ref == 0)


def test_relu_ties_route_evenly():
    """Post-ReLU activations: the minimum 0 is attained by many elements of many channels; the
    scale gradient must be split over all of them exactly like autograd does."""
    g = synth.gen(5)
    x = torch.relu(torch.randn(64, 96, generator=g))
    w = torch.randn(64, 96, generator=g)
    q = make_q({"n_bits": 4, "metric": "minmax", "method": "per_cluster", "cluster_dim": 1, "counts": [24] * 4})
    xg = x.to(DEV).requires_grad_(True)
    (dx,) = torch.autograd.grad((q(xg) * w.to(DEV)).sum(), [xg])
    xr = x.clone().requires_grad_(True)
    out, _ = O.act_forward(xr, O.ActState(), 4, "calibration", method="per_cluster", metric="minmax", cluster_dim=1,
                           cluster_counts=[24] * 4)
    (ref,) = torch.autograd.grad((out * w).sum(), [xr])
    assert rel_err(dx, ref) <= 1e-5


@pytest.mark.parametrize("shape,dt", [((2, 8, 64), "f32"), ((16, 64), "f16"), ((1, 2048, 4096), "f32"),
                                      ((2048, 11008), "f16"), ((1, 2048, 4096), "f16"), ((7, 33, 70), "f32"),
                                      ((64, 28672), "f32")])
def test_gather_bit_exact(shape, dt):
    x = synth.activation(shape, 3, DT[dt])
    idx = synth.permutation(shape[-1], 4)
    ref = O.gather_lastdim(x, idx)
    out = F.gather_lastdim(x.to(DEV), idx.to(torch.int32).to(DEV))
    assert same(out.cpu(), ref)
    # encode -> inverse gather round trip restores the input
    inv = torch.argsort(idx).to(torch.int32).to(DEV)
    assert same(F.gather_lastdim(out, inv).cpu(), x)


@pytest.mark.parametrize("shape,dt,n", [((1, 2048, 4096), "f16", 2), ((33, 70), "f32", 3), ((300, 1000), "f16", 8), ((5, 28672), "f32", 2)])
def test_batched_gather_equals_one_gather_per_tensor(shape, dt, n):
    """Several tensors of one shape, each through its own permutation, in one launch: the oracle's index_select per tensor,
    bit for bit (the residual-stream gathers of a block's two reordered norms at the bench shape; ragged shapes; 8 tensors)."""
    xs = [synth.activation(shape, 30 + i, DT[dt]) for i in range(n)]
    idxs = [synth.permutation(shape[-1], 40 + i) for i in range(n)]
    outs = F.gather_lastdim_batch([x.to(DEV) for x in xs], [i.to(torch.int32).to(DEV) for i in idxs])
    for x, i, o in zip(xs, idxs, outs):
        assert same(o.cpu(), O.gather_lastdim(x, i))
    from miq._lib import MiqError
    with pytest.raises(MiqError):
        F.gather_lastdim_batch([xs[0].to(DEV)] * 9, [idxs[0].to(torch.int32).to(DEV)] * 9)
    with pytest.raises(MiqError):
        F.gather_lastdim_batch([xs[0].to(DEV), xs[0][..., :8].contiguous().to(DEV)], [idxs[0].to(torch.int32).to(DEV)] * 2)


def test_gather_identity_and_wiring_fixture():
    d = golden("reorder")["wiring_small"]
    x3 = d["x3"].to(DEV)
    idx = d["ln1_reorder_index"].to(torch.int32).to(DEV)
    g = F.gather_lastdim(x3, idx)
    assert same(g.cpu(), torch.index_select(d["x3"], 2, d["ln1_reorder_index"]))
    ident = torch.arange(x3.shape[-1], dtype=torch.int32, device=DEV)
    assert same(F.gather_lastdim(x3, ident).cpu(), d["x3"])
