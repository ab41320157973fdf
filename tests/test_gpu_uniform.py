"""GPU parity of the uniform row-wise kernels (C ABI through ctypes) against the golden vectors
captured from the reference and against the CPU oracle at the benchmark sizes.

Parity contract.  With pre-activated clip factors (MIQ_LWC_PREACTIVATED: the test hands the kernel
the same sigmoid values the reference computed) and without LWC everything is a pure function of
IEEE ops and must be BIT-IDENTICAL to the reference: scale, zero point, integer codes, dequantized
values.  With the sigmoid fused into the kernel its last bit depends on the math library (ATen's CPU
sigmoid itself uses a 2-ulp exp), so:
  * per-row scale within 4 ulp of the reference, and bit-identical for >= 97% of the rows;
  * on rows whose (scale, zero point) are bit-identical: dequantized values and integer codes are
    bit-identical to the reference;
  * on the remaining rows: integer codes are bit-identical to what the reference's op sequence
    yields for the kernel's own (scale, zero point), and dequantized values are within 1 ulp... of
    (codes - zp) * scale evaluated with that scale (i.e. the only difference is the scale's last bit);
  * gradients: norm-wise relative error <= 1e-5 * max(1, qmax/15) in fp32 (the reference sums
    g*xd and g*m*x/s separately, so its own rounding error grows with qmax); 2e-2 in fp16 (the reference accumulates the row sums in fp16).
"""
import pytest
import torch

from helpers import DT, golden, rel_err, same, ulp_diff, weight_inputs
from miq import functional as F
from miq import synth
from oracle import miq_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rowify(t, rows):
    """[O, I] tensor -> [rows, I*O/rows] so that per-row masks apply (groups without padding only)."""
    return t.reshape(rows, -1)


def check_exact(out, scale, rzp, codes, ref_out, ref_scale, ref_rzp, ref_codes):
    assert same(scale.cpu(), ref_scale), "scale not bit-identical"
    assert same(rzp.cpu(), ref_rzp), "zero point not bit-identical"
    assert same(out.cpu(), ref_out), "dequantized values not bit-identical"
    if codes is not None:
        assert torch.equal(codes.cpu().to(torch.int32), torch.nan_to_num(ref_codes.float(), nan=0.0).to(torch.int32))


def check_forward(x, up, low, out, scale, rzp, codes, ref_out, ref_scale, ref_rzp, ref_codes, n_bits, group, lwc):
    out, scale, rzp = out.cpu(), scale.cpu(), rzp.cpu()
    codes = codes.cpu() if codes is not None else None
    assert scale.shape == ref_scale.shape
    ud = ulp_diff(scale, ref_scale).flatten()
    assert ud.max().item() <= 4, f"scale differs by {ud.max().item()} ulp"
    row_ok = (ud == 0) & (ulp_diff(rzp, ref_rzp).flatten() == 0)
    frac = row_ok.float().mean().item()
    assert frac >= 0.97, f"only {frac:.3f} of the rows have a bit-identical scale"
    rows = scale.numel()
    O_, I_ = x.shape
    padded = bool(group) and lwc and (I_ % group != 0)
    if padded:
        # per-(row, group) masks need the padded view; compare whole matrix when everything matched
        if bool(row_ok.all()):
            assert same(out, ref_out)
            if codes is not None:
                assert torch.equal(codes.to(torch.int32), ref_codes.to(torch.int32))
        return frac
    o2, r2 = _rowify(out, rows), _rowify(ref_out, rows)
    assert same(o2[row_ok], r2[row_ok]), "dequantized values differ on rows with identical scale"
    if codes is not None:
        c2, rc2 = _rowify(codes, rows).to(torch.int32), _rowify(ref_codes, rows).to(torch.int32)
        assert torch.equal(c2[row_ok], rc2[row_ok]), "codes differ on rows with identical scale"
    bad = ~row_ok
    if bool(bad.any()):
        xs = _rowify(x, rows)[bad]
        s, z = scale.reshape(rows, 1)[bad], rzp.reshape(rows, 1)[bad]
        q = xs / s
        lv = ((q.round() - q) + q).add(z).clamp(0, 2 ** n_bits - 1)
        if codes is not None:
            assert torch.equal(torch.nan_to_num(lv, nan=0.0).to(torch.int32), _rowify(codes, rows).to(torch.int32)[bad])
        assert same(lv.sub(z).mul(s), o2[bad])
    return frac


@pytest.mark.parametrize("name", sorted(golden("uniform").keys()))
def test_uniform_golden(name):
    d = golden("uniform")[name]
    m = d["meta"]
    x, up, low, w = weight_inputs(m)
    group, bits, sym, lwc = m["group_size"], m["n_bits"], m["symmetric"], m["lwc"]
    xd = x.to(DEV)
    upd = up.to(DEV).contiguous() if lwc else None
    lowd = low.to(DEV).contiguous() if lwc else None
    out, scale, rzp, codes = F.rowquant_uniform_fwd(xd, bits, sym, upd, lowd, group, want_codes=bits <= 8)
    if m["big"]:
        r = O.uniform_row(x, bits, sym, up, low, group or None)
        ref_out, ref_codes = r.out, r.codes
        assert synth.sha(ref_out) == d["out_sha"]  # oracle == reference on this case
    else:
        ref_out, ref_codes = d["out"], d["codes"]
    if lwc:
        check_forward(x, up, low, out, scale, rzp, codes, ref_out, d["scale"], d["rzp"], ref_codes, bits, group, lwc)
        su, sl = torch.sigmoid(up), torch.sigmoid(low)  # the reference's own clip factors
        o2, s2, z2, c2 = F.rowquant_uniform_fwd(xd, bits, sym, su.to(DEV), sl.to(DEV), group, want_codes=bits <= 8,
                                                preactivated=True)
        check_exact(o2, s2, z2, c2, ref_out, d["scale"], d["rzp"], ref_codes)
    else:
        check_exact(out, scale, rzp, codes, ref_out, d["scale"], d["rzp"], ref_codes)

    # backward
    wd = w.to(DEV)
    dx, d_up, d_low = F.rowquant_uniform_bwd(wd, xd, bits, sym, upd, lowd, group)
    tol = (1e-5 if x.dtype == torch.float32 else 2e-2) * max(1.0, (2 ** bits - 1) / 15)
    if m["big"]:
        xg = x.clone().requires_grad_(True)
        ps = [p.clone().requires_grad_(True) for p in (up, low)] if lwc else []
        r = O.uniform_row(xg, bits, sym, *(ps if lwc else (None, None)), group or None)
        g = torch.autograd.grad((r.out * w).sum(), [xg] + ps)
        ref_dx = g[0]
    else:
        ref_dx = d["dx"]
    # where the reference's own autograd overflows (fp16 s*s underflow, constant rows) it returns
    # NaN/inf; the kernel evaluates the closed form in fp32 and stays finite there: compare the rest.
    def masked(a, ref):
        ok = torch.isfinite(ref)
        z = torch.zeros_like(ref)
        return rel_err(torch.where(ok, a.cpu(), z), torch.where(ok, ref, z))

    if lwc:
        assert masked(d_up, d["d_up"]) <= tol, ("d_up", masked(d_up, d["d_up"]))
        assert masked(d_low, d["d_low"]) <= tol, ("d_low", masked(d_low, d["d_low"]))
        # pre-activated mode returns d/d(sigmoid): chain with sigmoid' on the host
        _, d_su, d_sl = F.rowquant_uniform_bwd(wd, xd, bits, sym, su.to(DEV), sl.to(DEV), group, want_dx=False,
                                               preactivated=True)
        assert masked(d_su.cpu().float() * (su * (1 - su)).float(), d["d_up"].float()) <= tol
        assert masked(d_sl.cpu().float() * (sl * (1 - sl)).float(), d["d_low"].float()) <= tol
    assert masked(dx, ref_dx) <= 4 * tol, masked(dx, ref_dx)


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
    out, scale, rzp, codes = F.rowquant_uniform_fwd(x.to(DEV), bits, sym, up.to(DEV), low.to(DEV), group, want_codes=True)
    check_forward(x, up, low, out, scale, rzp, codes, r.out, r.scale, r.rzp, r.codes, bits, group, True)
    su, sl = torch.sigmoid(up), torch.sigmoid(low)
    o2, s2, z2, c2 = F.rowquant_uniform_fwd(x.to(DEV), bits, sym, su.to(DEV), sl.to(DEV), group, want_codes=True,
                                            preactivated=True)
    check_exact(o2, s2, z2, c2, r.out, r.scale, r.rzp, r.codes)


def test_properties_full_size():
    """Size-independent properties on a Llama-2-7B MLP weight: idempotence of the level grid and
    codes/out consistency computed on the GPU itself."""
    x = synth.weight(11008, 4096, 5).to(DEV)
    up, low = (t.to(DEV) for t in synth.lwc_factors(11008, 6))
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
    dx, d_up, d_low = F.rowquant_uniform_bwd(g.to(DEV), x.to(DEV), 4, False, up.to(DEV), low.to(DEV))
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
