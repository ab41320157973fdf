"""Seeded random sweep of shapes, bit-widths and options through the row-wise and the activation kernels against the CPU
oracle: widths that are not a multiple of the 16-byte pack (the scalar tiles), one-row and one-column matrices, every
tile of the dispatch tables, groups that do not divide the row, cluster layouts with leftovers.  Forward results must be
the oracle's bits (the kernels take activated clip factors: the oracle's own sigmoid values are handed over)."""
import os
import random

import pytest
import torch

from helpers import DT, same
from miq import functional as F
from miq import synth
from oracle import miq_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

WIDTHS = [1, 3, 8, 17, 64, 100, 255, 256, 520, 768, 1000, 1024, 1028, 2048, 2052, 3072, 4100, 5120, 8192, 11008, 13824]


# other shapes and dtypes on request (a soak run over several offsets: MIQ_TEST_SEED_OFFSET=1 pytest tests/test_gpu_random_shapes.py)
SEED_OFFSET = int(os.environ.get("MIQ_TEST_SEED_OFFSET", "0"))


def _cases(n, seed):
    rnd = random.Random(seed + SEED_OFFSET)
    out = []
    for i in range(n):
        I = rnd.choice(WIDTHS)
        Orows = rnd.choice([1, 2, 5, 16, 33, 64]) if I > 4096 else rnd.choice([1, 2, 7, 64, 130, 257])
        dt = rnd.choice(["f32", "f16"])
        bits = rnd.choice([2, 3, 4, 5, 6, 8])
        sym = rnd.random() < 0.4
        lwc = rnd.random() < 0.6
        group = rnd.choice([0, 0, 0, 32, 64, 128]) if I >= 64 else 0
        if group and not lwc and (Orows * I) % group:
            group = 0  # without LWC the reference cuts the flat buffer into groups and needs it to divide
        out.append((i, Orows, I, dt, bits, sym, lwc, group))
    return out


@pytest.mark.parametrize("i,Orows,I,dt,bits,sym,lwc,group", _cases(48, 2024))
def test_uniform_forward_random_case(i, Orows, I, dt, bits, sym, lwc, group):
    dtype = DT[dt]
    x = synth.weight(Orows, I, 1000 + i, dtype)
    rows = Orows * (-(-I // group) if group else 1)
    up = low = su = sl = None
    if lwc:
        up, low = synth.lwc_factors(rows, 2000 + i, dtype)
        su, sl = torch.sigmoid(up).to(DEV), torch.sigmoid(low).to(DEV)
    # the oracle applies the sigmoid itself; the kernel receives the same values
    r = O.uniform_row(x, bits, sym, up, low, group or None)
    out, scale, rzp, codes = F.rowquant_uniform_fwd(x.to(DEV), bits, sym, su, sl, group, want_codes=True)
    assert same(scale.cpu(), r.scale), "scale"
    assert same(rzp.cpu(), r.rzp), "rounded zero point"
    assert same(out.cpu(), r.out), "dequantized values"
    assert torch.equal(codes.cpu().to(torch.int32), torch.nan_to_num(r.codes.float(), nan=0.0).to(torch.int32)), "codes"


def _act_cases(n, seed):
    rnd = random.Random(seed + SEED_OFFSET)
    out = []
    for i in range(n):
        C = rnd.choice([8, 64, 100, 256, 520, 1000, 1024, 2048, 2752, 4096, 5120])
        R = rnd.choice([1, 3, 16, 77, 300, 1025])
        dt = rnd.choice(["f32", "f16"])
        bits = rnd.choice([3, 4, 6, 8])
        kind = rnd.choice(["tensor", "equal", "ragged"])
        if kind == "tensor" or C < 16:
            counts = None
        elif kind == "equal":
            k = rnd.choice([d for d in (2, 4, 5, 8, 10, 16, 32) if C % d == 0] or [1])
            counts = [C // k] * k
        else:
            cuts = sorted(rnd.sample(range(1, C), min(5, C - 1)))
            counts = [b - a for a, b in zip([0] + cuts[:-1], cuts)]   # the tail past the last cut belongs to no cluster
        out.append((i, R, C, dt, bits, counts))
    return out


@pytest.mark.parametrize("i,R,C,dt,bits,counts", _act_cases(32, 77))
def test_activation_forward_random_case(i, R, C, dt, bits, counts):
    dtype = DT[dt]
    layout = F.ClusterLayout.get(counts, C, DEV)
    kw = dict(method="per_cluster" if counts else "per_tensor", metric="minmax", cluster_dim=1 if counts else None,
              cluster_counts=counts)
    x = synth.activation((R, C), 3000 + i, dtype)
    outs, codes, vecs = F.act_fresh_quant_fwd([x.to(DEV)], bits, layout, want_codes=True)
    ref, ref_codes = O.act_forward(x, O.ActState(), bits, "calibration", **kw)
    assert same(outs[0].cpu(), ref)
    assert torch.equal(codes[0].cpu().to(torch.int32), torch.nan_to_num(ref_codes, nan=0.0).to(torch.int32))


@pytest.mark.parametrize("i,R,C,dt,bits,counts", _act_cases(28, 99))
def test_activation_backward_random_case(i, R, C, dt, bits, counts):
    """The input gradient of the calibrating activation quantizer (two-launch form where the layout allows it, the
    four-kernel form otherwise) against the oracle's autograd: elements away from the cluster extremes to rounding, the
    few that receive a share of the scale gradient to the tolerance of another summation order.  Every other case has
    ReLU-style mass ties at its minimum."""
    from helpers import rel_err
    from miq import synth as S
    dtype = DT[dt]
    layout = F.ClusterLayout.get(counts, C, DEV)
    kw = dict(method="per_cluster" if counts else "per_tensor", metric="minmax", cluster_dim=1 if counts else None,
              cluster_counts=counts)
    x = S.activation((R, C), 3100 + i, dtype)
    if i % 2:
        x = torch.relu(x)
    g = S.grad_like(x, 3200 + i)
    xd, gd = x.to(DEV), g.to(DEV)
    _outs, _codes, vecs = F.act_fresh_quant_fwd([xd], bits, layout)
    dx = F.act_fresh_quant_bwd([gd], [xd], vecs, bits, layout)[0].cpu()
    xr = x.clone().requires_grad_(True)
    ref, _ = O.act_forward(xr, O.ActState(), bits, "calibration", **kw)
    (ref_dx,) = torch.autograd.grad((ref.float() * g.float()).sum(), [xr])
    ok = torch.isfinite(ref_dx)
    cmin, cmax = vecs[0][0].cpu(), vecs[0][1].cpu()
    routed = (x == cmax) | (x == cmin)
    z = torch.zeros_like(ref_dx)
    tight = 1e-5 if dtype == torch.float32 else 2e-2
    # routed elements carry the scale gradient, a sum over the whole cluster: the reference rounds every product to fp16
    # before it sums them (terms of both signs: up to a million per cluster here), the kernel accumulates in fp32
    loose = 5e-3 if dtype == torch.float32 else 5e-2
    for sel, tol in ((ok & ~routed, tight), (ok & routed, loose)):
        err = rel_err(torch.where(sel, dx, z), torch.where(sel, ref_dx, z))
        assert err <= tol, (i, err, tol)


def _row_cases(n, seed):
    rnd = random.Random(seed + SEED_OFFSET)
    out = []
    for i in range(n):
        I = rnd.choice(WIDTHS[2:])
        Orows = rnd.choice([1, 2, 5, 16, 33]) if I > 4096 else rnd.choice([1, 3, 64, 130])
        out.append((i, Orows, I, rnd.choice(["f32", "f16"]), rnd.choice([2, 3, 4]), rnd.random() < 0.5, rnd.random() < 0.6))
    return out


@pytest.mark.parametrize("i,Orows,I,dt,bits,sym,lwc", _row_cases(32, 31337))
def test_exponent_grids_forward_random_case(i, Orows, I, dt, bits, sym, lwc):
    from test_gpu_rowfamilies import check_exp_forward
    dtype = DT[dt]
    x = synth.weight(Orows, I, 4000 + i, dtype)
    up = low = su = sl = None
    if lwc:
        up, low = synth.lwc_factors(Orows, 5000 + i, dtype)
        su, sl = torch.sigmoid(up).to(DEV), torch.sigmoid(low).to(DEV)
    r2 = O.exp2_row(x, bits, up, low)
    out2, scale2 = F.rowquant_exp2_fwd(x.to(DEV), bits, su, sl)
    assert same(scale2.cpu(), r2.scale)
    check_exp_forward(x, r2.scale, out2, r2.out, 2)
    r1 = O.exp1_row(x, bits, sym, up, low)
    out1, scale1, _ = F.rowquant_exp1_fwd(x.to(DEV), bits, sym, su, sl)
    assert same(scale1.cpu(), r1.scale)
    check_exp_forward(x if sym else x - r1.rzp, r1.scale, out1, r1.out, 1)


@pytest.mark.parametrize("i,Orows,I,dt,bits,sym,lwc", _row_cases(24, 777))
def test_exponent_grids_backward_random_case(i, Orows, I, dt, bits, sym, lwc):
    """Input and clip-factor gradients of both exponent grids against the oracle's autograd (v2: the reference's own
    backward rule), whenever the forward has no element on the other side of a level threshold."""
    from test_gpu_rowfamilies import chain, check_exp_forward, masked_rel
    dtype = DT[dt]
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    x = synth.weight(Orows, I, 4100 + i, dtype)
    w = synth.grad_like(x, 4200 + i)
    up = low = su = sl = None
    ps = []
    if lwc:
        up, low = synth.lwc_factors(Orows, 5100 + i, dtype)
        su, sl = torch.sigmoid(up).to(DEV), torch.sigmoid(low).to(DEV)
        ps = [up.clone().requires_grad_(True), low.clone().requires_grad_(True)]
    for which in (2, 1):
        xg = x.clone().requires_grad_(True)
        args = ps if lwc else [None, None]
        r = O.exp2_row(xg, bits, *args) if which == 2 else O.exp1_row(xg, bits, sym, *args)
        g = torch.autograd.grad((r.out * w).sum(), [xg] + ps)
        if which == 2:
            out, _scale = F.rowquant_exp2_fwd(x.to(DEV), bits, su, sl)
            dx, d_su, d_sl = F.rowquant_exp2_bwd(w.to(DEV), x.to(DEV), bits, su, sl)
            ref_x = x
        else:
            out, _scale, _mid = F.rowquant_exp1_fwd(x.to(DEV), bits, sym, su, sl)
            dx, d_su, d_sl = F.rowquant_exp1_bwd(w.to(DEV), x.to(DEV), bits, sym, su, sl)
            ref_x = x if sym else x - r.rzp.detach()
        if check_exp_forward(ref_x, r.scale.detach(), out, r.out.detach(), which) != 0:
            # an element on the other side of a threshold changes its gradient by O(1): forward-only case.  Only fp16 rows
            # can have one (their level comes from the logarithm instruction); fp32 rows pick levels by exact thresholds
            assert dtype == torch.float16, (which, "an fp32 row with a level on the other side of a threshold")
            continue
        assert masked_rel(dx, g[0]) <= 4 * tol, (which, masked_rel(dx, g[0]))
        if lwc and dtype == torch.float32:  # (fp16: the reference's chain works in fp16 near its subnormal range there)
            # the clip-factor gradient is a row sum in which the contributions of the alive elements cancel (exactly in
            # the kernel's closed form, to rounding in the reference's chain): with one or two rows the reference's
            # residue shows (1.6e-4 seen), hence the looser bound than for the matrices of the golden cases
            # (and a residue of 1e-10 where the true gradient is exactly zero: an absolute floor of 1e-6 of sum |g| |x|)
            ftol = max(2 * tol, 5e-4)
            floor = 1e-6 * (w.float().abs() * x.float().abs()).sum(1, keepdim=True).norm()
            for d, sgm, ref in ((d_su, su, g[1]), (d_sl, sl, g[2])):
                err = (chain(d, sgm) - ref.float()).norm()
                assert err <= ftol * ref.float().norm() + floor, (which, float(err), float(ref.float().norm()))


@pytest.mark.parametrize("i,Orows,I,dt,bits,sym,lwc", [c for c in _row_cases(40, 4242) if c[3] == "f32" and c[2] % 4 == 0][:16])
def test_mixed_bit_and_let_forward_random_case(i, Orows, I, dt, bits, sym, lwc):
    g = synth.gen(6000 + i)
    x = synth.weight(Orows, I, 6000 + i)
    up = low = su = sl = None
    if lwc:
        up, low = synth.lwc_factors(Orows, 7000 + i)
        su, sl = torch.sigmoid(up).to(DEV), torch.sigmoid(low).to(DEV)
    nb = synth.mix_bits(Orows, 3.0, 8000 + i)
    for hard in (False, True):
        r = O.mix_row(x, nb.clone(), sym, up, low, None, hard, 0.0)
        out = F.rowquant_mix_fwd(x.to(DEV), nb.reshape(-1).to(DEV), sym, su, sl, 0, hard, 0.0)
        assert same(out.cpu(), r.out), ("mix", hard)
    if I <= 16384:
        cs = torch.rand(I, generator=g) * 2 + 0.25
        rs = torch.rand(Orows, generator=g) * 2 + 0.25
        for mode in ("div", "mul"):
            t = x * cs.view(1, -1)
            t = t / rs.view(-1, 1) if mode == "div" else t * rs.view(-1, 1)
            res = O.uniform_row(t, bits + 1, sym, up, low)
            out, _, scale, rzp = F.let_rowquant_fwd(x.to(DEV), cs.to(DEV), rs.to(DEV), mode == "div", None, su, sl, bits + 1, sym)
            assert same(scale.cpu(), res.scale) and same(rzp.cpu(), res.rzp) and same(out.cpu(), res.out), ("let", mode)


@pytest.mark.parametrize("i,Orows,I,dt,bits,sym,lwc", [c for c in _row_cases(48, 515) if c[3] == "f32" and c[2] % 4 == 0][:16])
def test_mixed_bit_backward_random_case(i, Orows, I, dt, bits, sym, lwc):
    """Gradients of the learnable mixed-bit quantizer (input, bit-width, clip factors) in its soft and hard phase against
    the oracle's autograd; rows with a whole-number bit-width, every row in the hard phase, take the single-grid path."""
    from test_gpu_rowfamilies import chain
    x = synth.weight(Orows, I, 6100 + i)
    w = synth.grad_like(x, 6200 + i)
    up = low = su = sl = None
    ps = []
    if lwc:
        up, low = synth.lwc_factors(Orows, 7100 + i)
        su, sl = torch.sigmoid(up).to(DEV), torch.sigmoid(low).to(DEV)
        ps = [up.clone().requires_grad_(True), low.clone().requires_grad_(True)]
    nb = synth.mix_bits(Orows, 3.0, 8100 + i)
    nb[::3] = nb[::3].round()  # whole-number rows in the soft phase too
    floor = 1e-6 * (w.abs() * x.abs()).sum(1, keepdim=True).norm()

    def close(a, ref, tol):
        return (a.float().cpu() - ref.float()).norm() <= tol * ref.float().norm() + floor
    for hard in (False, True):
        xg, ng = x.clone().requires_grad_(True), nb.clone().requires_grad_(True)
        r = O.mix_row(xg, ng, sym, *(ps if lwc else (None, None)), None, hard, 0.0)
        g = torch.autograd.grad((r.out * w).sum(), [xg, ng] + ps, allow_unused=True)
        dx, d_su, d_sl, d_n = F.rowquant_mix_bwd(w.to(DEV), x.to(DEV), nb.reshape(-1).to(DEV), sym, su, sl, 0, hard, 0.0)
        assert close(dx, g[0], 4e-5), ("dx", hard)
        if g[1] is not None:
            assert close(d_n.reshape(-1, 1), g[1], 5e-4), ("d_nbits", hard)
        if lwc:
            assert close(chain(d_su, su), g[2], 5e-4) and close(chain(d_sl, sl), g[3], 5e-4), ("clip factors", hard)


@pytest.mark.parametrize("I", [2048, 4096, 8192, 11008, 13824, 28672])
@pytest.mark.parametrize("dt", ["f32", "f16"])
def test_exponent_grids_backward_at_every_tile(I, dt):
    """The exponent grids' backward picks its tile from a table of its own for 2-byte elements: one case per tile of both
    tables, gradients against the oracle's autograd (fp16: the oracle accumulates its row sums in fp16, the kernel in fp32)."""
    from test_gpu_rowfamilies import chain, masked_rel
    dtype = DT[dt]
    Orows = 24
    x = synth.weight(Orows, I, 9000 + I % 97, dtype)
    up, low = synth.lwc_factors(Orows, 9100 + I % 97, dtype)
    w = synth.grad_like(x, 9200 + I % 97)
    su, sl = torch.sigmoid(up).to(DEV), torch.sigmoid(low).to(DEV)
    tol_x, tol_p = (4e-5, 2e-5) if dt == "f32" else (3e-2, 3e-2)
    for kind in (2, 1):
        xg = x.clone().requires_grad_(True)
        pu, pl = up.clone().requires_grad_(True), low.clone().requires_grad_(True)
        r = O.exp2_row(xg, 3, pu, pl) if kind == 2 else O.exp1_row(xg, 3, True, pu, pl)
        g = torch.autograd.grad((r.out * w).sum(), [xg, pu, pl])
        if kind == 2:
            out = F.rowquant_exp2_fwd(x.to(DEV), 3, su, sl)[0]
            dx, d_su, d_sl = F.rowquant_exp2_bwd(w.to(DEV), x.to(DEV), 3, su, sl)
        else:
            out = F.rowquant_exp1_fwd(x.to(DEV), 3, True, su, sl)[0]
            dx, d_su, d_sl = F.rowquant_exp1_bwd(w.to(DEV), x.to(DEV), 3, True, su, sl)
        # a level that flipped at a threshold changes the gradients of its ROW by construction (the element's own, the
        # row's scale gradient and with it the shares of the extremes): such rows are left out, every other row is checked
        ref_out = r.out.detach()
        rows_ok = ((out.cpu() == ref_out) | (torch.isnan(out.cpu()) & torch.isnan(ref_out))).all(dim=1)
        assert rows_ok.float().mean().item() >= 0.75, (kind, "levels flipped in too many rows", int((~rows_ok).sum()))
        assert masked_rel(dx.cpu()[rows_ok], g[0][rows_ok]) <= tol_x, (kind, masked_rel(dx.cpu()[rows_ok], g[0][rows_ok]))
        du, dl = chain(d_su, su), chain(d_sl, sl)
        assert masked_rel(du[rows_ok], g[1][rows_ok]) <= tol_p, (kind, "up", masked_rel(du[rows_ok], g[1][rows_ok]))
        assert masked_rel(dl[rows_ok], g[2][rows_ok]) <= tol_p, (kind, "low", masked_rel(dl[rows_ok], g[2][rows_ok]))
