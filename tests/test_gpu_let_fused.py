"""LET-fused weight quantizer (miq_let_rowquant_fwd / _bwd) against the op chain it replaces,
evaluated by the CPU oracle: t = (W * cs) / rs (or * rs) -> row-wise uniform fake-quant (+ LWC),
wshift = W @ shift, gradients by autograd through that chain.

Forward (quantized weight, scale, rounded zero point): bit-identical, with the clipping factors
handed over pre-activated (sigmoid evaluated once on the host, as in the other bit-exact tests).
wshift and all gradients are fp32 sums in a different order: norm-wise 1e-5 (2e-5 for the column sums).
"""
import pytest
import torch

from helpers import rel_err, same
from miq import functional as F
from miq import synth
from oracle import miq_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = [
    # O, I, cs, rs (None | "div" | "mul"), shift, lwc, symmetric, bits
    (64, 96, True, "div", True, True, False, 4),
    (64, 96, True, "mul", True, True, False, 4),
    (200, 256, True, None, True, True, False, 3),
    (200, 256, False, "div", False, False, False, 4),
    (130, 1000, True, "div", True, True, True, 4),
    (768, 768, True, "mul", True, True, False, 4),
    (3072, 768, True, None, True, True, False, 2),
    (1024, 4096, True, "div", True, True, False, 4),     # TEAM 256 x 4 packs, 512 blocks x 2 rows
    (600, 5120, True, "mul", True, True, False, 4),      # 256 x 6
    (300, 11008, True, None, True, True, False, 4),      # 512 x 6
    (300, 11008, True, "div", True, True, False, 4),
    (200, 8192, True, "div", True, True, False, 4),      # 256 x 8
    (100, 13824, True, "mul", True, True, False, 4),     # 512 x 8
    (4096, 4096, False, None, False, True, False, 4),    # plain quantizer through the fused kernels
]


def make(O_, I, cs, rs, shift, lwc, seed):
    g = synth.gen(seed)
    w = synth.weight(O_, I, seed)
    d = {"w": w}
    d["cs"] = (torch.rand(I, generator=g) * 2 + 0.25) if cs else None
    d["rs"] = (torch.rand(O_, generator=g) * 2 + 0.25) if rs else None
    d["shift"] = (torch.randn(I, generator=g) * 0.1) if shift else None
    if lwc:
        d["up"], d["low"] = synth.lwc_factors(O_, seed + 1)
    else:
        d["up"] = d["low"] = None
    d["g"] = synth.grad_like(w, seed + 2)
    d["gb"] = torch.randn(O_, generator=g)
    return d


@pytest.mark.parametrize("O_,I,cs,rs,shift,lwc,sym,bits", CASES)
def test_let_fused_matches_chain(O_, I, cs, rs, shift, lwc, sym, bits):
    d = make(O_, I, cs, rs, shift, lwc, 500 + O_ % 97)
    # ---- reference chain on the CPU
    leaf = {k: (d[k].clone().requires_grad_(True) if d[k] is not None else None) for k in ("cs", "rs", "shift", "up", "low")}
    t = d["w"]
    if cs:
        t = t * leaf["cs"].view(1, -1)
    if rs == "div":
        t = t / leaf["rs"].view(-1, 1)
    elif rs == "mul":
        t = t * leaf["rs"].view(-1, 1)
    res = O.uniform_row(t, bits, sym, leaf["up"], leaf["low"])
    loss = (res.out * d["g"]).sum()
    if shift:
        wshift_ref = d["w"] @ leaf["shift"]
        loss = loss + (wshift_ref * d["gb"]).sum()
    params = [v for v in leaf.values() if v is not None]
    grads = dict(zip([k for k, v in leaf.items() if v is not None], torch.autograd.grad(loss, params))) if params else {}
    # ---- fused kernels (they take the activated clipping factors)
    dev = lambda x: None if x is None else x.to(DEV)
    su = torch.sigmoid(d["up"]) if lwc else None
    sl = torch.sigmoid(d["low"]) if lwc else None
    out, wshift, scale, rzp = F.let_rowquant_fwd(dev(d["w"]), dev(d["cs"]), dev(d["rs"]), rs == "div", dev(d["shift"]),
                                                 dev(su), dev(sl), bits, sym)
    assert same(scale.cpu(), res.scale.detach()), "scale"
    assert same(rzp.cpu(), res.rzp.detach()), "rounded zero point"
    assert same(out.cpu(), res.out.detach()), "quantized weight"
    if shift:
        assert rel_err(wshift.cpu(), wshift_ref.detach()) <= (1e-6 if I <= 4096 else 3e-6)  # fp32 dot products, other order
    d_cs, d_rs, d_shift, d_su, d_sl = F.let_rowquant_bwd(dev(d["g"]), dev(d["gb"]) if shift else None, dev(d["w"]), dev(d["cs"]),
                                                         dev(d["rs"]), rs == "div", dev(d["shift"]), dev(su), dev(sl), bits, sym)
    if cs:
        assert rel_err(d_cs.cpu(), grads["cs"]) <= 2e-5, rel_err(d_cs.cpu(), grads["cs"])
    if rs:
        assert rel_err(d_rs.cpu(), grads["rs"]) <= 1e-5, rel_err(d_rs.cpu(), grads["rs"])
    if shift:
        assert rel_err(d_shift.cpu(), grads["shift"]) <= 2e-5
    if lwc:
        # chain rule through the sigmoid done here (the kernel returned gradients w.r.t. the activated factors)
        assert rel_err(d_su.cpu().reshape(-1) * (su * (1 - su)).reshape(-1), grads["up"].reshape(-1)) <= 1e-5
        assert rel_err(d_sl.cpu().reshape(-1) * (sl * (1 - sl)).reshape(-1), grads["low"].reshape(-1)) <= 1e-5


def test_autograd_function_and_fallback_predicate():
    d = make(256, 512, True, "div", True, True, 777)
    w = d["w"].to(DEV)
    d["up"], d["low"] = torch.sigmoid(d["up"]), torch.sigmoid(d["low"])  # both paths below take activated factors
    p = {k: d[k].to(DEV).requires_grad_(True) for k in ("cs", "rs", "shift", "up", "low")}
    out, wshift, scale, rzp = F.LetRowFQ.apply(w, p["cs"], p["rs"].view(-1, 1), True, p["shift"], p["up"], p["low"], 4, False)
    loss = (out * d["g"].to(DEV)).sum() + (wshift * d["gb"].to(DEV)).sum()
    got = torch.autograd.grad(loss, list(p.values()))
    # the same chain in eager PyTorch ops + the stand-alone quantizer kernels
    t = (w * p["cs"].view(1, -1)) / p["rs"].view(-1, 1)
    out2, _s, _z = F.UniformRowFQ.apply(t, p["up"], p["low"], 4, False, 0, None, False)
    loss2 = (out2 * d["g"].to(DEV)).sum() + ((w @ p["shift"]) * d["gb"].to(DEV)).sum()
    ref = torch.autograd.grad(loss2, list(p.values()))
    assert same(out.detach().cpu(), out2.detach().cpu())
    for a, b, k in zip(got, ref, p):
        assert a.shape == b.shape and rel_err(a.cpu(), b.cpu()) <= 2e-5, k
    assert F.let_supported(w) and not F.let_supported(w.half()) and not F.let_supported(w[:, :510]) and not F.let_supported(w, 128)


def test_half_output_and_half_gradient_are_casts():
    """out_dtype fp16 == the fp32 result cast to fp16; an fp16 incoming gradient gives exactly the
    gradients of the same values in fp32 (the kernels convert on load)."""
    d = make(512, 4096, True, "div", True, True, 901)
    dev = lambda x: None if x is None else x.to(DEV)
    args = (dev(d["w"]), dev(d["cs"]), dev(d["rs"]), True, dev(d["shift"]), dev(torch.sigmoid(d["up"])),
            dev(torch.sigmoid(d["low"])), 4, False)
    out32, ws32, s32, z32 = F.let_rowquant_fwd(*args)
    out16, ws16, s16, z16 = F.let_rowquant_fwd(*args, out_dtype=torch.float16)
    assert out16.dtype == torch.float16 and torch.equal(out16, out32.half())
    assert torch.equal(s16, s32) and torch.equal(z16, z32) and torch.equal(ws16, ws32)
    g16 = dev(d["g"]).half()
    a = F.let_rowquant_bwd(g16, dev(d["gb"]), *args[:3], True, *args[4:])
    b = F.let_rowquant_bwd(g16.float(), dev(d["gb"]), *args[:3], True, *args[4:])
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("dt", [torch.float32, torch.float16])
def test_truncate_inplace_matches_truncate_number(dt):
    """miq_truncate_inplace against the mirror of the reference's TruncateFunction (models/transformation.py:4-20),
    bit for bit: zeros stay zero, NaN stays NaN, more vectors than one launch takes."""
    from miq.let import truncate_number
    g = torch.Generator(device="cuda:0").manual_seed(4)
    vecs = []
    for i in range(19):
        v = (torch.randn(37 + 300 * i, generator=g, device="cuda:0") * 0.02).to(dt)
        v[::7] = 0
        v[3] = float("nan")
        v[5] = 0.01
        v[6] = -0.0099
        vecs.append(v)
    vecs.append(torch.empty(0, device="cuda:0", dtype=dt))
    want = [truncate_number(v.clone()) for v in vecs]
    F.truncate_inplace_(vecs, 1e-2)
    for a, b in zip(vecs, want):
        assert torch.equal(torch.nan_to_num(a, nan=7.0), torch.nan_to_num(b, nan=7.0))
        assert torch.equal(torch.signbit(a), torch.signbit(b))


@pytest.mark.parametrize("rs", [None, "div", "mul"])
def test_let_fused_forward_on_degenerate_rows(rs):
    """Rows of zeros, constants, an infinity, a NaN, subnormal and huge values: a row whose scale is zero, infinite or NaN
    takes the plain division (chosen per row), its neighbours the rounded-level product; all must give the chain's bits."""
    g = synth.gen(901)
    O_, I = 12, 1024
    w = synth.weight(O_, I, 902)
    w[0] = 0.0
    w[1] = 0.25
    w[2, 9] = float("inf")
    w[3, 11] = float("nan")
    w[4] = torch.randn(I, generator=g).sign() * 1e-42
    w[5] = torch.randn(I, generator=g).sign() * 3e38
    w[6, ::2] = 0.0
    w[7, 5] = -float("inf")
    cs = torch.rand(I, generator=g) * 2 + 0.25
    r = (torch.rand(O_, generator=g) * 2 + 0.25) if rs else None
    shift = torch.randn(I, generator=g) * 0.1
    for sym in (False, True):
        t = w * cs.view(1, -1)
        if rs == "div":
            t = t / r.view(-1, 1)
        elif rs == "mul":
            t = t * r.view(-1, 1)
        res = O.uniform_row(t, 4, sym)
        dev = lambda x: None if x is None else x.to(DEV)
        out, wshift, scale, rzp = F.let_rowquant_fwd(dev(w), dev(cs), dev(r), rs == "div", dev(shift), None, None, 4, sym)
        assert same(scale.cpu(), res.scale), "scale"
        assert same(rzp.cpu(), res.rzp), "rounded zero point"
        assert same(out.cpu(), res.out), "quantized weight"
        out16 = F.let_rowquant_fwd(dev(w), dev(cs), dev(r), rs == "div", dev(shift), None, None, 4, sym, out_dtype=torch.float16)[0]
        assert same(out16.cpu(), res.out.half()), "fp16 store"
