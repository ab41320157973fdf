"""GPU parity of the exponential (v1, v2), learnable mixed-bit and fixed mixed-bit row-wise
quantizers and of the reorder-index kernel against the golden vectors captured from the reference.

Contract:
  * fixed / learnable mixed-bit forward: pure IEEE arithmetic -> BIT-IDENTICAL.  The kernels take activated clip
    factors; kernel-level checks hand over the reference's own (CPU sigmoid, as in the golden run), module-level checks
    let the module evaluate torch.sigmoid on the GPU and compare with the oracle for those clip factors;
  * exponential quantizers: the level index comes from log2f.  Outputs must be bit-identical except
    for elements whose |x/scale| lies within 8 ulp of a level threshold (0.5 and 1.5*2^k for v2,
    2^(k+0.5) for v1), where the neighbouring level is accepted; such elements are counted and must be
    fewer than 1e-4 of the tensor;
  * gradients: norm-wise relative error <= 1e-5 (fp32) / 2e-2 (fp16) on entries where the
    reference itself is finite;
  * reorder permutations: bit-identical.
"""
import numpy as np
import pytest
import torch

from helpers import DT, golden, gpu_sigmoid, rel_err, same, weight_inputs
from miq import functional as F
from miq import synth
from miq.quantizer_exp import ExpAffineQuantizer as Exp1Q
from miq.quantizer_exp_v2 import ExpAffineQuantizer as Exp2Q
from miq.quantizer_mix import MixAffineQuantizer
from miq.quantizer_mix_v2 import Mix2AffineQuantizer
from oracle import miq_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def masked_rel(a, ref):
    ok = torch.isfinite(ref)
    z = torch.zeros_like(ref)
    return rel_err(torch.where(ok, a.detach().cpu(), z), torch.where(ok, ref, z))


def sig(up, low):
    return (torch.sigmoid(up).to(DEV), torch.sigmoid(low).to(DEV)) if up is not None else (None, None)


def chain(d_s, s):
    """gradient w.r.t. the raw factor from the gradient w.r.t. its sigmoid"""
    s = s.float().cpu()
    return d_s.float().cpu() * s * (1 - s)


def near_threshold(x, scale, kind):
    """mask of elements whose |x/scale| is within 8 ulp (of the tensor dtype) of a level threshold"""
    q = (x / scale).abs().double()
    eps = 8 * (2.0 ** -23 if x.dtype == torch.float32 else 2.0 ** -10)
    if kind == 2:
        th = torch.tensor([0.5, 0.75] + [1.5 * 2.0 ** k for k in range(0, 16)], dtype=torch.float64)
    else:
        th = torch.tensor([2.0 ** (k + 0.5) for k in range(-20, 2)], dtype=torch.float64)
    rel = ((q.unsqueeze(-1) - th) / th).abs().min(-1).values
    return rel <= eps


def check_exp_forward(x, scale_ref, out, ref_out, kind):
    out = out.cpu()
    if same(out, ref_out):
        return 0
    diff = ~((out == ref_out) | (torch.isnan(out) & torch.isnan(ref_out)))
    ok = near_threshold(x, scale_ref, kind)
    assert not bool((diff & ~ok).any()), f"{int((diff & ~ok).sum())} elements differ away from any level threshold"
    assert diff.float().mean().item() < 1e-4
    return int(diff.sum())


# ------------------------------------------------------------------------------------ fixed mixed-bit
@pytest.mark.parametrize("name", sorted(golden("mix2").keys()))
def test_mix2_golden(name):
    d = golden("mix2")[name]
    m = d["meta"]
    x, up, low, w = weight_inputs(m)
    bits = d["bits"].to(DEV)
    su, sl = sig(up, low)
    out, scale, rzp, _ = F.rowquant_uniform_fwd(x.to(DEV), 0, m["symmetric"], su, sl, 0, bits, True)
    assert same(out.cpu(), d["out"])
    # drop-in module (evaluates torch.sigmoid on the GPU): forward bit-identical to the reference op sequence for those
    # clip factors, gradients to tolerance
    q = Mix2AffineQuantizer(target_bit=3, symmetric=m["symmetric"], per_channel_axes=[0], dynamic=True,
                            dynamic_method="per_channel", shape=tuple(m["shape"]), mix2_nbits=d["bits"], lwc=m["lwc"]).to(DEV)
    if m["lwc"]:
        with torch.no_grad():
            q.upbound_factor.copy_(up)
            q.lowbound_factor.copy_(low)
    xg = x.to(DEV).requires_grad_(True)
    y = q(xg)
    params = [q.upbound_factor, q.lowbound_factor] if m["lwc"] else []
    g = torch.autograd.grad((y * w.to(DEV)).sum(), [xg] + params)
    sug, slg = gpu_sigmoid(up, low)
    r = O.uniform_row(x, d["bits"].unsqueeze(-1), m["symmetric"], sug, slg, None, clamp_asym_scale=True, activated=True)
    assert same(y.detach().cpu(), r.out)
    assert masked_rel(y, d["out"]) <= 1e-5
    assert masked_rel(g[0], d["dx"]) <= 4e-5
    if m["lwc"]:
        assert masked_rel(g[1], d["d_up"]) <= 1e-5 and masked_rel(g[2], d["d_low"]) <= 1e-5


# ------------------------------------------------------------------------------------ exp v2
@pytest.mark.parametrize("name", sorted(golden("exp2").keys()))
def test_exp2_golden(name):
    d = golden("exp2")[name]
    m = d["meta"]
    x, up, low, w = weight_inputs(m)
    tol = 1e-5 if x.dtype == torch.float32 else 2e-2
    su, sl = sig(up, low)
    out, scale = F.rowquant_exp2_fwd(x.to(DEV), m["n_bits"], su, sl)
    assert same(scale.cpu(), d["scale"])
    if m["big"]:
        r = O.exp2_row(x, m["n_bits"], up, low)
        assert synth.sha(r.out) == d["out_sha"]
        ref_out = r.out
        xg = x.clone().requires_grad_(True)
        ps = [p.clone().requires_grad_(True) for p in (up, low)] if m["lwc"] else []
        r = O.exp2_row(xg, m["n_bits"], *(ps if m["lwc"] else (None, None)))
        ref_dx = torch.autograd.grad((r.out * w).sum(), [xg] + ps)[0]
    else:
        ref_out, ref_dx = d["out"], d["dx"]
    nflip = check_exp_forward(x, d["scale"], out, ref_out, 2)
    dx, d_su, d_sl = F.rowquant_exp2_bwd(w.to(DEV), x.to(DEV), m["n_bits"], su, sl)
    if nflip == 0:
        assert masked_rel(dx, ref_dx) <= 4 * tol, masked_rel(dx, ref_dx)
        if m["lwc"]:
            assert masked_rel(chain(d_su, su), d["d_up"].float()) <= tol
            assert masked_rel(chain(d_sl, sl), d["d_low"].float()) <= tol
    # drop-in module (torch.sigmoid on the GPU, then the kernel)
    q = Exp2Q(n_bits=m["n_bits"], symmetric=True, per_channel_axes=[0], dynamic=True, dynamic_method="per_channel",
              shape=tuple(m["shape"]), lwc=m["lwc"])
    q = (q.half() if x.dtype == torch.float16 else q).to(DEV)
    if m["lwc"]:
        with torch.no_grad():
            q.upbound_factor.copy_(up)
            q.lowbound_factor.copy_(low)
    xg = x.to(DEV).requires_grad_(True)
    y = q(xg)
    (y.float() * w.to(DEV).float()).sum().backward()
    assert xg.grad is not None and (not m["lwc"] or q.upbound_factor.grad is not None)
    sug, slg = gpu_sigmoid(up, low)
    r = O.exp2_row(x, m["n_bits"], sug, slg, activated=True)
    assert same(q.scale.detach().cpu(), r.scale)
    check_exp_forward(x, r.scale, y.detach(), r.out, 2)
    if m["lwc"] and nflip == 0 and torch.equal(sug, torch.sigmoid(up)) and torch.equal(slg, torch.sigmoid(low)):
        assert masked_rel(q.upbound_factor.grad, d["d_up"]) <= tol and masked_rel(q.lowbound_factor.grad, d["d_low"]) <= tol


# ------------------------------------------------------------------------------------ exp v1
@pytest.mark.parametrize("name", sorted(golden("exp1").keys()))
def test_exp1_golden(name):
    d = golden("exp1")[name]
    m = d["meta"]
    x, up, low, w = weight_inputs(m)
    tol = 1e-5 if x.dtype == torch.float32 else 2e-2
    su, sl = sig(up, low)
    out, scale, mid = F.rowquant_exp1_fwd(x.to(DEV), m["n_bits"], m["symmetric"], su, sl)
    assert same(scale.cpu(), d["scale"])
    if "mid" in d:
        assert same(mid.cpu(), d["mid"])
    xs = x if "mid" not in d else x - d["mid"]
    nflip = check_exp_forward(xs, d["scale"], out, d["out"], 1)
    dx, d_su, d_sl = F.rowquant_exp1_bwd(w.to(DEV), x.to(DEV), m["n_bits"], m["symmetric"], su, sl)
    if nflip == 0:
        assert masked_rel(dx, d["dx"]) <= 4 * tol, masked_rel(dx, d["dx"])
        if m["lwc"]:
            assert masked_rel(chain(d_su, su), d["d_up"].float()) <= tol, masked_rel(chain(d_su, su), d["d_up"].float())
            assert masked_rel(chain(d_sl, sl), d["d_low"].float()) <= tol
    q = Exp1Q(n_bits=m["n_bits"], symmetric=m["symmetric"], per_channel_axes=[0], dynamic=True,
              dynamic_method="per_channel", shape=tuple(m["shape"]), lwc=m["lwc"])
    q = (q.half() if x.dtype == torch.float16 else q).to(DEV)
    if m["lwc"]:
        with torch.no_grad():
            q.upbound_factor.copy_(up)
            q.lowbound_factor.copy_(low)
    y = q(x.to(DEV))
    assert y.shape == x.shape and q.scale.shape == d["scale"].shape
    sug, slg = gpu_sigmoid(up, low)
    r = O.exp1_row(x, m["n_bits"], m["symmetric"], sug, slg, activated=True)
    assert same(q.scale.detach().cpu(), r.scale)
    check_exp_forward(x if r.rzp is None else x - r.rzp, r.scale, y.detach(), r.out, 1)


# ------------------------------------------------------------------------------------ learnable mixed-bit
@pytest.mark.parametrize("name", sorted(golden("mix").keys()))
def test_mix_golden(name):
    d = golden("mix")[name]
    m = d["meta"]
    x, up, low, w = weight_inputs(m)
    su, sl = sig(up, low)
    nb = d["n_bits_after"].reshape(-1).to(DEV)
    out = F.rowquant_mix_fwd(x.to(DEV), nb, m["symmetric"], su, sl, 0, m["hard"], m["threshold"])
    assert same(out.cpu(), d["out"])
    dx, d_su, d_sl, d_n = F.rowquant_mix_bwd(w.to(DEV), x.to(DEV), nb, m["symmetric"], su, sl, 0, m["hard"], m["threshold"])
    assert masked_rel(dx, d["dx"]) <= 4e-5, masked_rel(dx, d["dx"])
    assert masked_rel(d_n.reshape(-1, 1), d["d_nbits"]) <= 2e-5, masked_rel(d_n.reshape(-1, 1), d["d_nbits"])
    if m["lwc"]:
        assert masked_rel(chain(d_su, su), d["d_up"]) <= 1e-5
        assert masked_rel(chain(d_sl, sl), d["d_low"]) <= 1e-5
    # drop-in module: clamps n_bits in place, exposes the size loss
    q = MixAffineQuantizer(target_bit=3, n_bits=d["n_bits_in"].clone(), hard_assign=m["hard"], threshold=m["threshold"],
                           symmetric=m["symmetric"], per_channel_axes=[0], dynamic=True, dynamic_method="per_channel",
                           shape=tuple(m["shape"]), lwc=m["lwc"]).to(DEV)
    if m["lwc"]:
        with torch.no_grad():
            q.upbound_factor.copy_(up)
            q.lowbound_factor.copy_(low)
    xg = x.to(DEV).requires_grad_(True)
    y = q(xg)
    assert same(q.n_bits.detach().cpu(), d["n_bits_after"])
    sug, slg = gpu_sigmoid(up, low)
    r = O.mix_row(x, d["n_bits_after"], m["symmetric"], sug, slg, None, m["hard"], m["threshold"], activated=True)
    assert same(y.detach().cpu(), r.out)
    (y * w.to(DEV)).sum().backward()
    agree = not m["lwc"] or (torch.equal(sug, torch.sigmoid(up)) and torch.equal(slg, torch.sigmoid(low)))
    # (where the GPU's sigmoid differs from the CPU's in a last bit, that row's scale, hence a few of its codes, differ
    # from the golden run's)
    assert masked_rel(q.n_bits.grad, d["d_nbits"]) <= (2e-5 if agree else 1e-3)
    ms, ts = q.model_size_loss()
    # plain torch reductions (summation order differs between devices): equal to rounding
    assert torch.allclose(ms.detach().cpu(), d["model_size"], rtol=1e-6) and same(ts.detach().cpu(), d["target_size"])
    q.set_hard()
    q.round_nbits()
    assert not isinstance(q.n_bits, torch.nn.Parameter)
    assert torch.equal(q.n_bits, q.n_bits.round())


# ------------------------------------------------------------------------------------ reorder index
@pytest.mark.parametrize("name", sorted(k for k in golden("reorder") if k.startswith("r_")))
def test_reorder_index_golden(name):
    d = golden("reorder")[name]
    m = d["meta"]
    idx = F.reorder_index(d["xmax"].to(DEV), d["xmin"].to(DEV), m["n_heads"] or 1)
    assert torch.equal(idx.cpu(), d["index"])


def test_reorder_index_ties_are_stable_and_large():
    g = synth.gen(9)
    r = torch.randint(0, 50, (11008,), generator=g).float()  # many ties
    idx = F.reorder_index(r.to(DEV), torch.zeros_like(r).to(DEV), 1).cpu()
    assert torch.equal(idx, torch.argsort(r, stable=True))
    r = torch.rand(28672, generator=g)
    idx = F.reorder_index(r.to(DEV), torch.zeros_like(r).to(DEV), 64).cpu()
    ref, _ = O.reorder_index(r, torch.zeros_like(r), 4, 64)
    assert torch.equal(idx, ref)
    assert torch.equal(torch.sort(idx).values, torch.arange(28672))


# ------------------------------------------------------------------------------------ benchmark shapes vs the oracle
@pytest.mark.parametrize("shape,bits", [((4096, 4096), 3), ((11008, 4096), 3), ((4096, 11008), 4), ((1024, 28672), 2)])
def test_exp2_llama_shapes_vs_oracle(shape, bits):
    """Llama-2-7B weight shapes (the W3A16 exponential-quantizer configuration, fwd + STE bwd)."""
    x = synth.weight(shape[0], shape[1], 2)
    up, low = synth.lwc_factors(shape[0], 3)
    w = synth.grad_like(x, 4)
    xg = x.clone().requires_grad_(True)
    pu, pl = up.clone().requires_grad_(True), low.clone().requires_grad_(True)
    r = O.exp2_row(xg, bits, pu, pl)
    g = torch.autograd.grad((r.out * w).sum(), [xg, pu, pl])
    su, sl = sig(up, low)
    out, scale = F.rowquant_exp2_fwd(x.to(DEV), bits, su, sl)
    assert same(scale.cpu(), r.scale.detach())
    nflip = check_exp_forward(x, r.scale.detach(), out, r.out.detach(), 2)
    dx, d_su, d_sl = F.rowquant_exp2_bwd(w.to(DEV), x.to(DEV), bits, su, sl)
    if nflip == 0:
        assert masked_rel(dx, g[0]) <= 4e-5
        assert masked_rel(chain(d_su, su), g[1]) <= 2e-5 and masked_rel(chain(d_sl, sl), g[2]) <= 2e-5


@pytest.mark.parametrize("shape,bits,symmetric", [((4096, 4096), 3, True), ((4096, 4096), 4, False), ((2048, 11008), 3, True)])
def test_exp1_llama_shapes_vs_oracle(shape, bits, symmetric):
    """The first exponential quantizer (quantize/quantizer_exp.py:92-166) at Llama-2-7B weight shapes, fwd + backward."""
    x = synth.weight(shape[0], shape[1], 12)
    up, low = synth.lwc_factors(shape[0], 13)
    w = synth.grad_like(x, 14)
    xg = x.clone().requires_grad_(True)
    pu, pl = up.clone().requires_grad_(True), low.clone().requires_grad_(True)
    r = O.exp1_row(xg, bits, symmetric, pu, pl)
    g = torch.autograd.grad((r.out * w).sum(), [xg, pu, pl])
    su, sl = sig(up, low)
    out, scale, mid = F.rowquant_exp1_fwd(x.to(DEV), bits, symmetric, su, sl)
    assert same(scale.cpu(), r.scale.detach())
    xs = x
    if not symmetric:
        assert same(mid.cpu(), r.rzp.detach())
        xs = x - r.rzp.detach()
    nflip = check_exp_forward(xs, r.scale.detach(), out, r.out.detach(), 1)
    dx, d_su, d_sl = F.rowquant_exp1_bwd(w.to(DEV), x.to(DEV), bits, symmetric, su, sl)
    if nflip == 0:
        assert masked_rel(dx, g[0]) <= 4e-5
        assert masked_rel(chain(d_su, su), g[1]) <= 2e-5 and masked_rel(chain(d_sl, sl), g[2]) <= 2e-5


@pytest.mark.parametrize("shape,hard", [((5120, 5120), False), ((13824, 5120), True), ((1280, 13824), False)])
def test_mix_llama13b_shapes_vs_oracle(shape, hard):
    """Llama-2-13B weight shapes (the learnable per-output-channel bit-width configuration), soft and hard phase."""
    x = synth.weight(shape[0], shape[1], 5)
    up, low = synth.lwc_factors(shape[0], 6)
    nb = synth.mix_bits(shape[0], 3.0, 7)
    w = synth.grad_like(x, 8)
    xg = x.clone().requires_grad_(True)
    pu, pl, pn = up.clone().requires_grad_(True), low.clone().requires_grad_(True), nb.clone().requires_grad_(True)
    r = O.mix_row(xg, pn, False, pu, pl, None, hard, 0.0)
    g = torch.autograd.grad((r.out * w).sum(), [xg, pn, pu, pl])
    su, sl = sig(up, low)
    nbd = nb.reshape(-1).to(DEV)
    out = F.rowquant_mix_fwd(x.to(DEV), nbd, False, su, sl, 0, hard, 0.0)
    assert same(out.cpu(), r.out.detach())
    dx, d_su, d_sl, d_n = F.rowquant_mix_bwd(w.to(DEV), x.to(DEV), nbd, False, su, sl, 0, hard, 0.0)
    assert masked_rel(dx, g[0]) <= 4e-5
    assert masked_rel(d_n.reshape(-1, 1), g[1]) <= 4e-5
    assert masked_rel(chain(d_su, su), g[2]) <= 2e-5 and masked_rel(chain(d_sl, sl), g[3]) <= 2e-5


def test_mix2_llama13b_shape_vs_oracle():
    x = synth.weight(5120, 13824, 9)
    bits = synth.mix2_bits(5120, 10)
    up, low = synth.lwc_factors(5120, 11)
    r = O.uniform_row(x, bits.unsqueeze(-1), False, up, low, None, clamp_asym_scale=True)
    su, sl = sig(up, low)
    out, scale, rzp, codes = F.rowquant_uniform_fwd(x.to(DEV), 0, False, su, sl, 0, bits.to(DEV), True, want_codes=True)
    assert same(out.cpu(), r.out) and same(scale.cpu(), r.scale) and same(rzp.cpu(), r.rzp)
    assert torch.equal(codes.cpu().float(), r.codes)
    # rows never use more levels than their bit-width allows
    assert bool((codes.cpu().float().amax(1) <= 2 ** bits - 1).all())


def test_row_sharded_weight_quantization_equals_full():
    """Output-channel shards (the multi-GPU layout) need no exchange: quantizing shards == quantizing the matrix."""
    x = synth.weight(4096, 4096, 12).to(DEV)
    up, low = (torch.sigmoid(t).to(DEV) for t in synth.lwc_factors(4096, 13))
    full, s_full, z_full, _ = F.rowquant_uniform_fwd(x, 4, False, up, low)
    parts = []
    for r in range(8):
        a, b = r * 512, (r + 1) * 512
        o, s, z, _ = F.rowquant_uniform_fwd(x[a:b].contiguous(), 4, False, up[a:b].contiguous(), low[a:b].contiguous())
        parts.append(o)
    assert torch.equal(torch.cat(parts), full)


@pytest.mark.parametrize("dt", [torch.float32, torch.float16])
def test_exponent_grids_on_degenerate_rows(dt):
    """Zero, constant, infinite, NaN, subnormal and huge rows through both exponent quantizers: the scale is clamped, so
    every row takes the corrected product; the forward must still give the reference's bits on each of them."""
    g = torch.Generator().manual_seed(12)
    width = 1024
    big = 6.0e4 if dt == torch.float16 else 3.0e38
    tiny = 6.0e-8 if dt == torch.float16 else 1.0e-42
    x = torch.randn(8, width, generator=g) * 0.02
    x[0] = 0.0
    x[1] = 0.5
    x[2, 5] = float("inf")
    x[3, 7] = float("nan")
    x[4] = torch.randn(width, generator=g).sign() * tiny
    x[5] = torch.randn(width, generator=g).sign() * big
    x[6, ::2] = 0.0
    x = x.to(dt)
    for bits in (2, 3, 4):
        r = O.exp2_row(x, bits)
        out, scale = F.rowquant_exp2_fwd(x.to(DEV), bits)
        assert same(scale.cpu(), r.scale)
        check_exp_forward(x, r.scale, out, r.out, 2)
        for symmetric in (True, False):
            r1 = O.exp1_row(x, bits, symmetric)
            out1, scale1, mid1 = F.rowquant_exp1_fwd(x.to(DEV), bits, symmetric)
            assert same(scale1.cpu(), r1.scale)
            check_exp_forward(x, r1.scale, out1, r1.out, 1)


@pytest.mark.parametrize("hard", [False, True])
def test_mixed_bit_grids_on_degenerate_rows(hard):
    """The learnable and the fixed mixed-bit quantizer on rows of zeros, constants, infinities, NaN, subnormal and huge
    values, with bit-widths 2 to 8 spread over the rows: forward bits of the reference on every row."""
    g = torch.Generator().manual_seed(13)
    width = 1024
    x = torch.randn(16, width, generator=g) * 0.02
    x[0] = 0.0
    x[1] = 0.5
    x[2, 5] = float("inf")
    x[3, 7] = float("nan")
    x[4] = torch.randn(width, generator=g).sign() * 1e-42
    x[5] = torch.randn(width, generator=g).sign() * 3e38
    x[6, ::2] = 0.0
    x[7, 3] = -float("inf")
    nb = torch.linspace(2.0, 8.0, 16).view(-1, 1) + (0.0 if hard else 0.37)
    nb = nb.clamp(2, 8)
    for symmetric in (False, True):
        r = O.mix_row(x, nb.clone(), symmetric, None, None, None, hard, 0.0)
        out = F.rowquant_mix_fwd(x.to(DEV), nb.reshape(-1).to(DEV), symmetric, None, None, 0, hard, 0.0)
        assert same(out.cpu(), r.out), ("learnable", symmetric)
    bits = (torch.arange(16) % 7 + 2).float()
    for symmetric in (False, True):
        r2 = O.uniform_row(x, bits.unsqueeze(-1), symmetric, None, None, None, clamp_asym_scale=True)
        out2, scale2, rzp2, _ = F.rowquant_uniform_fwd(x.to(DEV), 0, symmetric, None, None, 0, bits.to(DEV), True)
        assert same(scale2.cpu(), r2.scale) and same(rzp2.cpu(), r2.rzp), ("fixed", symmetric)
        assert same(out2.cpu(), r2.out), ("fixed", symmetric)


def _ulp_neighbours(center, n):
    import numpy as np
    c = np.float32(center)
    out, lo, hi = [c], c, c
    for _ in range(n):
        lo = np.nextafter(lo, np.float32(-np.inf))
        hi = np.nextafter(hi, np.float32(np.inf))
        out += [lo, hi]
    return [float(v) for v in out]


@pytest.mark.parametrize("bits", [2, 3, 4])
def test_exp2_fp32_is_exact_at_every_level_threshold(bits):
    """fp32 rows with scale exactly 1 (row maximum = 2 * top), so that |x / s| = |x|: every float within 6 ulp of every
    level threshold (0.5, 0.75, 1.5 * 2^k up to the top level), both signs.  The kernel's comparisons and exponent-field
    rule must give what the reference's float arithmetic gives there (torch's CPU log2), bit for bit: no element may
    differ, near a threshold or not."""
    top = float(1 << ((1 << (bits - 1)) - 2))
    probes = []
    for c in [0.5, 0.75] + [1.5 * 2.0 ** k for k in range(0, 8) if 1.5 * 2.0 ** k < 4 * top]:
        probes += _ulp_neighbours(c, 6)
    probes = [p for p in probes if p < 2 * top]
    row = torch.tensor(probes + [-p for p in probes], dtype=torch.float32)
    n = 8 * ((row.numel() + 1 + 7) // 8)
    x = torch.zeros(4, n, dtype=torch.float32)
    x[:, :row.numel()] = row
    x[:, -1] = 2 * top                     # the row maximum: s = 2 * top / (2 * top) = 1
    x[1] = x[1].flip(0).roll(3)            # the same probes in other lanes
    x[1, -1], x[1, 0] = 2 * top, x[1, -1]
    x[2] *= -1                             # the extreme on the negative side
    x[3, :row.numel()] = row * 0.5         # thresholds of a scale that is not 1 ...
    x[3, -1] = top                         # ... s = 0.5
    ref = O.exp2_row(x, bits)
    out, scale = F.rowquant_exp2_fwd(x.to(DEV), bits)
    assert same(scale.cpu().reshape(-1), ref.scale.reshape(-1))
    assert torch.equal(out.cpu(), ref.out), (out.cpu() != ref.out).nonzero()[:8]
    g = torch.ones_like(x)
    xr = x.clone().requires_grad_(True)
    (ref_dx,) = torch.autograd.grad((O.exp2_row(xr, bits).out * g).sum(), [xr])
    dx, _, _ = F.rowquant_exp2_bwd(g.to(DEV), x.to(DEV), bits)
    ext = (x == x.amax(1, keepdim=True)) | (x == x.amin(1, keepdim=True))  # (receive the routed scale gradient)
    assert torch.allclose(dx.cpu()[~ext], ref_dx[~ext], rtol=2e-6, atol=0), "alive / dead sets differ at a threshold"


@pytest.mark.parametrize("bits", [2, 3, 4])  # the entry points take 2 to 4 bits
def test_exp1_fp32_is_exact_at_every_level_threshold(bits):
    """The same for the v1 grid (symmetric: scale = row maximum = 1): every float within 6 ulp of 2^(k - 0.5) for every
    level boundary k = e_lo .. 1, both signs, forward bit for bit and the live set of the backward."""
    e_lo = -((1 << (bits - 1)) - 1)
    probes = []
    for k in range(max(e_lo, -16) - 1, 2):
        probes += _ulp_neighbours(2.0 ** (k - 0.5), 6)
    probes = [p for p in probes if p < 1.0]
    row = torch.tensor(probes + [-p for p in probes], dtype=torch.float32)
    n = 8 * ((row.numel() + 1 + 7) // 8)
    x = torch.zeros(3, n, dtype=torch.float32)
    x[:, :row.numel()] = row
    x[:, -1] = 1.0
    x[1] = x[1].flip(0).roll(5)
    x[1, -1], x[1, 0] = 1.0, x[1, -1]
    x[2] *= -1
    ref = O.exp1_row(x, bits, True)
    out, scale, _mid = F.rowquant_exp1_fwd(x.to(DEV), bits, True)
    assert same(scale.cpu().reshape(-1), ref.scale.reshape(-1))
    assert torch.equal(out.cpu(), ref.out), (out.cpu() != ref.out).nonzero()[:8]
    g = torch.ones_like(x)
    xr = x.clone().requires_grad_(True)
    (ref_dx,) = torch.autograd.grad((O.exp1_row(xr, bits, True).out * g).sum(), [xr])
    dx, _, _ = F.rowquant_exp1_bwd(g.to(DEV), x.to(DEV), bits, True)
    ext = (x.abs() == x.abs().amax(1, keepdim=True))
    assert torch.allclose(dx.cpu()[~ext], ref_dx[~ext], rtol=2e-6, atol=0), "live sets differ at a threshold"
