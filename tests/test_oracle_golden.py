"""Pins the CPU oracle (oracle/miq_oracle.py) to golden vectors captured from the reference.

Forward results (dequantized values, scales, zero points, integer codes, indices) must be
bit-identical.  Gradients come from autograd through the same op sequence and are required to be
bit-identical as well (both sides are torch CPU kernels executing the same graph).
"""
import numpy as np
import pytest
import torch

from helpers import DT, golden, same, weight_inputs
from miq import synth
from oracle import miq_oracle as O


def _grads(out, w, tensors):
    loss = (out.float() * w.float()).sum() if out.dtype == torch.float16 else (out * w).sum()
    return torch.autograd.grad(loss, tensors, allow_unused=True)


def _check_big(d, name, t):
    assert synth.sha(t) == d[name + "_sha"], name
    assert same(t.detach()[d["rows"]], d[name + "_rows"]), name


@pytest.mark.parametrize("name", sorted(golden("uniform").keys()))
def test_uniform(name):
    d = golden("uniform")[name]
    m = d["meta"]
    x, up, low, w = weight_inputs(m)
    assert synth.sha(x) == d["x_sha"]
    xg = x.clone().requires_grad_(True)
    ps = [p.clone().requires_grad_(True) for p in (up, low)] if m["lwc"] else [None, None]
    r = O.uniform_row(xg, m["n_bits"], m["symmetric"], ps[0], ps[1], m["group_size"] or None)
    assert same(r.scale.detach(), d["scale"])
    assert same(r.rzp.detach(), d["rzp"])
    g = _grads(r.out, w, [xg] + [p for p in ps if p is not None])
    if m["big"]:
        _check_big(d, "out", r.out.detach())
        _check_big(d, "codes", r.codes.to(torch.uint8))
        _check_big(d, "dx", g[0])
    else:
        assert same(r.out.detach(), d["out"])
        assert torch.equal(torch.nan_to_num(r.codes, nan=0.0).to(torch.int32), d["codes"])
        assert same(g[0], d["dx"])
    if m["lwc"]:
        assert same(g[1], d["d_up"]) and same(g[2], d["d_low"])


@pytest.mark.parametrize("name", sorted(golden("mix2").keys()))
def test_mix2(name):
    d = golden("mix2")[name]
    m = d["meta"]
    x, up, low, w = weight_inputs(m)
    bits = synth.mix2_bits(m["shape"][0], m["seed"] + 3)
    assert torch.equal(bits, d["bits"])
    xg = x.clone().requires_grad_(True)
    ps = [p.clone().requires_grad_(True) for p in (up, low)] if m["lwc"] else [None, None]
    r = O.uniform_row(xg, bits.unsqueeze(-1), m["symmetric"], ps[0], ps[1], None, clamp_asym_scale=True)
    assert same(r.out.detach(), d["out"])
    g = _grads(r.out, w, [xg] + [p for p in ps if p is not None])
    assert same(g[0], d["dx"])
    if m["lwc"]:
        assert same(g[1], d["d_up"]) and same(g[2], d["d_low"])


@pytest.mark.parametrize("name", sorted(golden("exp1").keys()))
def test_exp1(name):
    d = golden("exp1")[name]
    m = d["meta"]
    x, up, low, w = weight_inputs(m)
    xg = x.clone().requires_grad_(True)
    ps = [p.clone().requires_grad_(True) for p in (up, low)] if m["lwc"] else [None, None]
    r = O.exp1_row(xg, m["n_bits"], m["symmetric"], ps[0], ps[1])
    assert same(r.scale.detach(), d["scale"])
    if "mid" in d:
        assert same(r.rzp.detach(), d["mid"])
    assert same(r.out.detach(), d["out"])
    g = _grads(r.out, w, [xg] + [p for p in ps if p is not None])
    assert same(g[0], d["dx"])
    if m["lwc"]:
        assert same(g[1], d["d_up"]) and same(g[2], d["d_low"])


@pytest.mark.parametrize("name", sorted(golden("exp2").keys()))
def test_exp2(name):
    d = golden("exp2")[name]
    m = d["meta"]
    x, up, low, w = weight_inputs(m)
    xg = x.clone().requires_grad_(True)
    ps = [p.clone().requires_grad_(True) for p in (up, low)] if m["lwc"] else [None, None]
    r = O.exp2_row(xg, m["n_bits"], ps[0], ps[1])
    assert same(r.scale.detach(), d["scale"])
    g = _grads(r.out, w, [xg] + [p for p in ps if p is not None])
    if m["big"]:
        _check_big(d, "out", r.out.detach())
        _check_big(d, "dx", g[0])
    else:
        assert same(r.out.detach(), d["out"])
        assert same(g[0], d["dx"])
    if m["lwc"]:
        assert same(g[1], d["d_up"]) and same(g[2], d["d_low"])


@pytest.mark.parametrize("name", sorted(golden("mix").keys()))
def test_mix(name):
    d = golden("mix")[name]
    m = d["meta"]
    x, up, low, w = weight_inputs(m)
    nb = d["n_bits_in"].clone().clamp(2, 8).requires_grad_(True)
    assert same(nb.detach(), d["n_bits_after"])
    xg = x.clone().requires_grad_(True)
    ps = [p.clone().requires_grad_(True) for p in (up, low)] if m["lwc"] else [None, None]
    r = O.mix_row(xg, nb, m["symmetric"], ps[0], ps[1], m["group_size"] or None, m["hard"], m["threshold"])
    assert same(r.out.detach(), d["out"])
    g = _grads(r.out, w, [xg, nb] + [p for p in ps if p is not None])
    assert same(g[0], d["dx"])
    assert same(g[1], d["d_nbits"])
    if m["lwc"]:
        assert same(g[2], d["d_up"]) and same(g[3], d["d_low"])
    ms, ts = O.mix_model_size(nb, m["shape"], 3, m["hard"], m["threshold"])
    assert same(ms.detach(), d["model_size"]) and same(ts.detach(), d["target_size"])


@pytest.mark.parametrize("name", sorted(golden("act").keys()))
def test_act(name):
    d = golden("act")[name]
    m = d["meta"]
    dtype, shape, seed = DT[m["dtype"]], tuple(m["shape"]), m["seed"]
    kw = dict(method=m["method"], metric=m["metric"],
              cluster_dim=None if m["cluster_dim"] < 0 else m["cluster_dim"],
              cluster_counts=m["counts"] or None)
    st = O.ActState()
    for c in range(m["calls"]):
        x = synth.activation(shape, seed + c, dtype)
        if c == 0:
            xg = x.clone().requires_grad_(True)
            out, _ = O.act_forward(xg, O.ActState(), m["n_bits"], "calibration", **kw)
            w = synth.grad_like(x, seed + 100)
            (dx,) = torch.autograd.grad((out.float() * w.float()).sum(), [xg])
            assert same(dx, d["dx0"])
        with torch.no_grad():
            out, _ = O.act_forward(x, st, m["n_bits"], "calibration", **kw)
        assert same(out, d[f"out{c}"]), c
        assert same(st.scale, d[f"scale{c}"]) and same(st.zero_point, d[f"zp{c}"]) and same(st.rzp, d[f"rzp{c}"])
        assert same(st.cached_min, d[f"cmin{c}"]) and same(st.cached_max, d[f"cmax{c}"])
    x = synth.activation(shape, seed + 77, dtype)
    xg = x.clone().requires_grad_(True)
    out, _ = O.act_forward(xg, st, m["n_bits"], "eval")
    w = synth.grad_like(x, seed + 101)
    (dx,) = torch.autograd.grad((out.float() * w.float()).sum(), [xg])
    assert same(out.detach(), d["out_eval"]) and same(dx, d["dx_eval"])


@pytest.mark.parametrize("name", sorted(k for k in golden("reorder") if k.startswith("r_")))
def test_reorder_index(name):
    d = golden("reorder")[name]
    m = d["meta"]
    if m["as_list"]:
        idx, counts = O.reorder_index([d["xmax"], d["kmax"]], [d["xmin"], d["kmin"]], m["n_clusters"], m["n_heads"] or None)
    else:
        idx, counts = O.reorder_index(d["xmax"], d["xmin"], m["n_clusters"], m["n_heads"] or None)
    assert torch.equal(idx, d["index"])
    assert np.array_equal(np.asarray(counts), d["counts"].numpy())


@pytest.mark.parametrize("name", sorted(k for k in golden("reorder") if k.startswith("h_")))
def test_reorder_stats(name):
    d = golden("reorder")[name]
    m = d["meta"]
    acc = None
    for k, s in enumerate(m["shapes"]):
        t = synth.activation(tuple(s), m["seed"] + k, DT[m["dtype"]])
        acc = O.running_minmax(acc, O.channel_minmax(t))
    assert same(acc[0], d["max"]) and same(acc[1], d["min"])


def test_gather_matches_wiring_fixture():
    d = golden("reorder")["wiring_small"]
    x3 = d["x3"]
    g = O.gather_lastdim(x3, d["ln1_reorder_index"])
    ref = torch.nn.functional.layer_norm(g, (x3.shape[-1],), d["post_ln1_w"], d["post_ln1_b"], 1e-5)
    assert same(ref, d["ln1_out3"])


# --------------------------------------------------------------------------------------------
# activation scale / shift statistics (fixture produced by the reference's get_act_scales /
# get_act_shifts) and the bit allocation rule
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["f32", "f16"])
def test_act_stats_oracle_matches_reference_functions(name):
    from helpers import StatToy
    d = golden("act_stats")[name]
    m = d["meta"]
    dt = DT[m["dtype"]]
    k = StatToy().k
    scales, shifts = {}, {}
    for i in range(m["n"]):
        x = synth.activation(tuple(m["shape"]), m["seed0"] + i, dt)
        for key, inp in (("a", x), ("b", torch.relu(x) * 0.5 + 1.0), ("c", -x[..., :k])):
            scales[key] = O.act_scale_update(scales.get(key), inp)
            shifts[key] = O.act_shift_update(shifts.get(key), inp)
    for key in ("a", "b", "c"):
        assert same(scales[key], d["scales"][key]) and same(shifts[key], d["shifts"][key]), key


def _fq_input(m):
    shape, dtype = tuple(m["shape"]), DT[m["dtype"]]
    if m["kind"] == "rows":
        return synth.weight(shape[0], shape[1], m["seed"] + 1, dtype)
    return synth.activation(shape, m["seed"], dtype)


@pytest.mark.parametrize("name", sorted(golden("fake_quant").keys()))
def test_fake_quant_with_given_parameters(name):
    d = golden("fake_quant")[name]
    m = d["meta"]
    out = O.fake_quant_given(_fq_input(m), d["scale"], d["rzp"], m["n_bits"], m["group_size"] or None, m["deficiency"])
    assert same(out, d["out"])


def test_allocate_bits_rule():
    s = torch.tensor([0.5, 0.1, 0.9, 0.3, 0.7, 0.2, 0.8, 0.4, 0.6, 0.05])
    assert O.allocate_bits(s).tolist() == [3, 3, 4, 3, 3, 3, 3, 3, 3, 2]
