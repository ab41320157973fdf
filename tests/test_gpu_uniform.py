"""GPU parity of the uniform row-wise kernels (C ABI through ctypes) and of the UniformAffineQuantizer module that
calls them, against the golden vectors captured from the reference and against the CPU oracle at the benchmark sizes.

Parity contract.  The library takes ACTIVATED clip factors (sigmoid(upbound_factor), sigmoid(lowbound_factor): the
tensors the reference forms at quantize/quantizer.py:259-261), so everything it computes is a pure function of IEEE ops
and must be BIT-IDENTICAL to the reference: scale, zero point, integer codes, dequantized values.
  * kernel level: handed the reference's own clip factors (torch.sigmoid on the CPU, as in the golden run), the kernels
    must reproduce the golden vectors bit for bit;
  * module level: the module evaluates torch.sigmoid on the GPU -- the routine the reference itself would run there --
    and must be bit-identical to the reference op sequence for those clip factors (the oracle with the GPU's
    sigmoid values); on every row where the GPU's and the CPU's sigmoid agree (in practice all of them) that is the
    golden vector itself;
  * gradients: norm-wise relative error <= 1e-5 * max(1, qmax/15) in fp32 (the reference sums g*xd and g*m*x/s
    separately, so its own rounding error grows with qmax); 2e-2 in fp16 (the reference accumulates the row sums in
    fp16).
"""
import pytest
import torch

from helpers import DT, golden, gpu_sigmoid, rel_err, same, weight_inputs
from miq import functional as F
from miq import synth
from miq.quantizer import UniformAffineQuantizer
from oracle import miq_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rowify(t, rows):
    """[O, I] tensor -> [rows, I*O/rows] so that per-row masks apply (groups without padding only)."""
    return t.reshape(rows, -1)


def check_exact(out, scale, rzp, codes, ref_out, ref_scale, ref_rzp, ref_codes):
    assert same(scale.detach().cpu().reshape(ref_scale.shape), ref_scale), "scale not bit-identical"
    assert same(rzp.detach().cpu().reshape(ref_rzp.shape), ref_rzp), "zero point not bit-identical"
    assert same(out.detach().cpu(), ref_out), "dequantized values not bit-identical"
    if codes is not None:
        assert torch.equal(codes.cpu().to(torch.int32), torch.nan_to_num(ref_codes.float(), nan=0.0).to(torch.int32))


def masked(a, ref):
    """rel_err where the reference is finite: where its own autograd overflows (fp16 s*s underflow, constant rows) it
    returns NaN / inf; the kernel evaluates the closed form in fp32 and stays finite there."""
    ok = torch.isfinite(ref)
    z = torch.zeros_like(ref)
    return rel_err(torch.where(ok, a.detach().cpu().to(ref.dtype), z), torch.where(ok, ref, z))


def module_for(x, bits, sym, group, lwc, up=None, low=None):
    """The drop-in module as a QuantLinear builds it for a weight (quantize/int_linear.py:47-58), on the GPU."""
    q = UniformAffineQuantizer(n_bits=bits, symmetric=sym, per_channel_axes=[0], metric="minmax", dynamic=True,
                               dynamic_method="per_channel", group_size=group or None, shape=tuple(x.shape), lwc=lwc)
    q = (q.half() if x.dtype == torch.float16 else q).to(DEV)
    if lwc:
        with torch.no_grad():
            q.upbound_factor.copy_(up)
            q.lowbound_factor.copy_(low)
    return q


def check_module(x, w, bits, sym, group, lwc, up, low, golden_case=None, tol=None):
    """forward() of the module on the GPU: bit-identical to the reference op sequence (scale, zero point, dequantized
    values); gradients through the module's own autograd path (ATen sigmoid backward included) to tolerance."""
    q = module_for(x, bits, sym, group, lwc, up, low)
    xg = x.to(DEV).requires_grad_(True)
    y = q(xg)
    su, sl = gpu_sigmoid(up, low) if lwc else (None, None)
    r = O.uniform_row(x, bits, sym, su, sl, group or None, activated=True)
    check_exact(y, q.scale, q.round_zero_point, None, r.out, r.scale, r.rzp, None)
    if golden_case is not None and lwc and not golden_case["meta"]["big"]:
        # rows whose clip factors the GPU's sigmoid and the CPU's agree on are the golden rows themselves
        agree = (su == torch.sigmoid(up)) & (sl == torch.sigmoid(low))
        if bool(agree.all()):
            check_exact(y, q.scale, q.round_zero_point, None, golden_case["out"], golden_case["scale"], golden_case["rzp"], None)
        else:
            rows = agree.numel()
            if y.numel() % rows == 0 and not (group and x.shape[1] % group):
                ok = agree.flatten()
                assert same(_rowify(y.detach().cpu(), rows)[ok], _rowify(golden_case["out"], rows)[ok])
        assert agree.float().mean().item() >= 0.9, "the device sigmoid disagrees with the CPU's on more than a few rows"
    if w is not None:
        (y.float() * w.to(DEV).float()).sum().backward()
        return q, xg
    return q, xg


@pytest.mark.parametrize("name", sorted(golden("uniform").keys()))
def test_uniform_golden(name):
    d = golden("uniform")[name]
    m = d["meta"]
    x, up, low, w = weight_inputs(m)
    group, bits, sym, lwc = m["group_size"], m["n_bits"], m["symmetric"], m["lwc"]
    xd = x.to(DEV)
    su, sl = (torch.sigmoid(up), torch.sigmoid(low)) if lwc else (None, None)  # the reference's own clip factors
    sud = su.to(DEV).contiguous() if lwc else None
    sld = sl.to(DEV).contiguous() if lwc else None
    out, scale, rzp, codes = F.rowquant_uniform_fwd(xd, bits, sym, sud, sld, group, want_codes=bits <= 8)
    if m["big"]:
        r = O.uniform_row(x, bits, sym, up, low, group or None)
        ref_out, ref_codes = r.out, r.codes
        assert synth.sha(ref_out) == d["out_sha"]  # oracle == reference on this case
    else:
        ref_out, ref_codes = d["out"], d["codes"]
    check_exact(out, scale, rzp, codes, ref_out, d["scale"], d["rzp"], ref_codes)

    # backward
    wd = w.to(DEV)
    dx, d_su, d_sl = F.rowquant_uniform_bwd(wd, xd, bits, sym, sud, sld, group)
    tol = (1e-5 if x.dtype == torch.float32 else 2e-2) * max(1.0, (2 ** bits - 1) / 15)
    if m["big"]:
        xg = x.clone().requires_grad_(True)
        ps = [p.clone().requires_grad_(True) for p in (up, low)] if lwc else []
        r = O.uniform_row(xg, bits, sym, *(ps if lwc else (None, None)), group or None)
        g = torch.autograd.grad((r.out * w).sum(), [xg] + ps)
        ref_dx = g[0]
    else:
        ref_dx = d["dx"]
    if lwc:
        # the kernel returns d / d(activated factor): chain with sigmoid' on the host
        assert masked(d_su.cpu().float() * (su * (1 - su)).float(), d["d_up"].float()) <= tol
        assert masked(d_sl.cpu().float() * (sl * (1 - sl)).float(), d["d_low"].float()) <= tol
        _, d_su2, d_sl2 = F.rowquant_uniform_bwd(wd, xd, bits, sym, sud, sld, group, want_dx=False)
        assert same(d_su2.cpu(), d_su.cpu()) and same(d_sl2.cpu(), d_sl.cpu())
    assert masked(dx, ref_dx) <= 4 * tol, masked(dx, ref_dx)

    # the module's own call path: forward bit-exact, gradients through ATen's sigmoid backward
    q, xg = check_module(x, w, bits, sym, group, lwc, up, low, golden_case=d)
    assert masked(xg.grad, ref_dx) <= 4 * tol, masked(xg.grad, ref_dx)
    if lwc:
        assert masked(q.upbound_factor.grad, d["d_up"]) <= tol, masked(q.upbound_factor.grad, d["d_up"])
        assert masked(q.lowbound_factor.grad, d["d_low"]) <= tol, masked(q.lowbound_factor.grad, d["d_low"])


@pytest.mark.parametrize("shape,dt,bits,sym,group", [
    ((4096, 4096), "f32", 4, False, 0),
    ((11008, 4096), "f32", 4, False, 0),
    ((4096, 11008), "f32", 4, False, 0),
    ((4096, 11008), "f16", 4, False, 0),
    ((4096, 4096), "f16", 3, True, 128),
    ((1024, 28672), "f32", 4, False, 0),
    ((1024, 28672), "f16", 8, False, 0),
    ((512, 5120), "f32", 2, True, 0),
    ((512, 13824), "f32", 6, False, 0),
])
def test_uniform_vs_oracle_benchmark_shapes(shape, dt, bits, sym, group):
    dtype = DT[dt]
    x = synth.weight(shape[0], shape[1], 2, dtype)
    rows = shape[0] * (-(-shape[1] // group) if group else 1)
    up, low = synth.lwc_factors(rows, 3, dtype)
    r = O.uniform_row(x, bits, sym, up, low, group or None)
    su, sl = torch.sigmoid(up), torch.sigmoid(low)
    o2, s2, z2, c2 = F.rowquant_uniform_fwd(x.to(DEV), bits, sym, su.to(DEV), sl.to(DEV), group, want_codes=True)
    check_exact(o2, s2, z2, c2, r.out, r.scale, r.rzp, r.codes)
    check_module(x, None, bits, sym, group, True, up, low)


def test_properties_full_size():
    """Size-independent properties on a Llama-2-7B MLP weight: idempotence of the level grid and
    codes/out consistency computed on the GPU itself."""
    x = synth.weight(11008, 4096, 5).to(DEV)
    up, low = (torch.sigmoid(t).to(DEV) for t in synth.lwc_factors(11008, 6))
    out, scale, rzp, codes = F.rowquant_uniform_fwd(x, 4, False, up, low, 0, want_codes=True)
    assert int(codes.max()) <= 15
    deq = (codes.float() - rzp) * scale
    assert torch.equal(deq, out)
    # every row uses at most 16 distinct values
    assert all(out[i].unique().numel() <= 16 for i in range(0, 11008, 997))
    # re-quantizing the output with the same (unclipped) grid reproduces it: levels are fixed points
    lv = torch.clamp(torch.round(out / scale) + rzp, 0, 15)
    assert torch.equal(lv.to(torch.uint8), codes)


def test_empty_and_errors():
    x = torch.empty((0, 64), device=DEV)
    out, scale, rzp, _ = F.rowquant_uniform_fwd(x, 4, False)
    assert out.shape == (0, 64) and scale.shape == (0, 1)
    from miq._lib import MiqError
    with pytest.raises(MiqError):
        F.rowquant_uniform_fwd(torch.zeros(4, 8), 4, False)  # CPU tensor: no fallback
    with pytest.raises(MiqError):
        F.rowquant_uniform_fwd(torch.zeros(4, 8, device=DEV), 1, False)  # n_bits out of range
    with pytest.raises(MiqError):
        F.rowquant_uniform_fwd(torch.zeros(4, 8, device=DEV, dtype=torch.bfloat16), 4, False)


@pytest.mark.parametrize("I,dt", [(5120, "f32"), (11008, "f32"), (13824, "f32"), (28672, "f32"), (28672, "f16"), (11008, "f16"),
                                  (2048, "f32"), (1540, "f32"), (4096, "f16"), (3080, "f16"), (8192, "f16")])  # incl. the 2-pack tile
def test_backward_at_llama_row_widths(I, dt):
    """Row widths that leave part of the widest register tiles idle (5120 ... 28672): gradients vs the oracle."""
    dtype = DT[dt]
    O_ = 48
    x = synth.weight(O_, I, 40, dtype)
    up, low = synth.lwc_factors(O_, 41, dtype)
    g = synth.grad_like(x, 42)
    xr, ur, lr = (t.clone().requires_grad_(True) for t in (x, up, low))
    r = O.uniform_row(xr, 4, False, ur, lr)
    ref = torch.autograd.grad((r.out.float() * g.float()).sum(), [xr, ur, lr])
    su, sl = torch.sigmoid(up), torch.sigmoid(low)
    dx, d_su, d_sl = F.rowquant_uniform_bwd(g.to(DEV), x.to(DEV), 4, False, su.to(DEV), sl.to(DEV))
    d_up = d_su.cpu().float() * (su * (1 - su)).float()
    d_low = d_sl.cpu().float() * (sl * (1 - sl)).float()
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for a, b, name in zip((dx, d_up, d_low), ref, ("dx", "d_up", "d_low")):
        ok = torch.isfinite(b)
        z = torch.zeros_like(b)
        assert rel_err(torch.where(ok, a.cpu(), z).float(), torch.where(ok, b, z).float()) <= 4 * tol, name


def _degenerate_rows(dtype, width):
    """Rows that leave the usual range of a weight row: zeros, constants, an infinity, a NaN, subnormal and huge values."""
    g = torch.Generator().manual_seed(11)
    big = 6.0e4 if dtype == torch.float16 else 3.0e38
    tiny = 6.0e-8 if dtype == torch.float16 else 1.0e-42
    x = torch.randn(10, width, generator=g) * 0.02
    x[0] = 0.0
    x[1] = 0.5
    x[2] = -3.0
    x[3, 5] = float("inf")
    x[4, 7] = float("nan")
    x[5] = torch.randn(width, generator=g).sign() * tiny
    x[6] = torch.randn(width, generator=g).sign() * big
    x[7, ::2] = 0.0          # half zeros
    x[8, 3] = -float("inf")
    x = x.to(dtype)          # row 9 stays an ordinary row
    return x


@pytest.mark.parametrize("dt", ["f32", "f16"])
@pytest.mark.parametrize("sym", [False, True])
@pytest.mark.parametrize("width", [512, 4096])
def test_degenerate_rows_match_the_oracle(dt, sym, width):
    """Rows whose scale is zero, infinite or NaN (an asymmetric scale is not clamped) take the kernels' general quotient,
    chosen once per row; ordinary rows beside them take the corrected product.  Both must give the reference's bits."""
    x = _degenerate_rows(DT[dt], width)
    for bits in (4, 8):
        r = O.uniform_row(x, bits, sym)
        out, scale, rzp, codes = F.rowquant_uniform_fwd(x.to(DEV), bits, sym, want_codes=True)
        check_exact(out, scale, rzp, None, r.out, r.scale, r.rzp, None)
        outs, scales, rzps = F.rowquant_uniform_fwd_batch([x.to(DEV), x.flip(0).contiguous().to(DEV)], bits, sym)
        assert same(outs[0].cpu(), r.out) and same(outs[1].cpu(), r.out.flip(0))
        assert same(scales[0].cpu(), r.scale) and same(rzps[0].cpu(), r.rzp)


@pytest.mark.parametrize("name", sorted(golden("fake_quant").keys()))
def test_public_fake_quant_matches_the_reference(name):
    """UniformAffineQuantizer.fake_quant(x, scale, round_zero_point) called on its own (quantize/quantizer.py:99-121):
    row-wise parameters of a grouped weight, including rows that are zero-padded to a multiple of the group size
    (deficiency > 0: pad, quantize, strip), row-wise without groups, a per-channel vector and a single scale.  Bit-identical
    to the reference's method on the same tensors."""
    d = golden("fake_quant")[name]
    m = d["meta"]
    shape, dtype = tuple(m["shape"]), DT[m["dtype"]]
    if m["kind"] == "rows":
        x = synth.weight(shape[0], shape[1], m["seed"] + 1, dtype)
        q = UniformAffineQuantizer(n_bits=m["n_bits"], symmetric=True, per_channel_axes=[0], dynamic=True,
                                   dynamic_method="per_channel", group_size=m["group_size"] or None, shape=shape, lwc=True)
        assert q.deficiency == m["deficiency"]
    else:
        x = synth.activation(shape, m["seed"], dtype)
        q = UniformAffineQuantizer(n_bits=m["n_bits"], dynamic=False, dynamic_method="per_cluster")
    out = q.fake_quant(x.to(DEV), d["scale"].to(DEV), d["rzp"].to(DEV))
    assert out.shape == x.shape
    assert same(out.cpu(), d["out"])
    assert same(O.fake_quant_given(x, d["scale"], d["rzp"], m["n_bits"], m["group_size"] or None, m["deficiency"]), d["out"])
