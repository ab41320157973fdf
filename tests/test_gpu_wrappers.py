"""Drop-in wrappers on the GPU: R1..R5 rewiring, statistics hooks, OmniLayerNorm gather,
QuantLinear / QuantMatMul forward, compared with fixtures captured from the reference's own
module objects (tests/golden/reorder.pt: wiring_small, h_*)."""
import types

import numpy as np
import pytest
import torch

from helpers import DT, golden, rel_err, same
from miq import reorder_utils as R
from miq import synth
from miq.int_linear import QuantLinear
from miq.int_matmul import QuantMatMul
from miq.omni_norm import OmniLayerNorm

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

WQ = {"n_bits": 4, "per_channel_axes": [0], "symmetric": False, "dynamic_method": "per_channel", "group_size": None,
      "lwc": True, "dynamic": True, "metric": "minmax", "target_bit": 4}
AQ = {"n_bits": 4, "per_channel_axes": [], "symmetric": False, "dynamic_method": "per_cluster", "dynamic": False,
      "metric": "minmax"}


def build(d):
    H, Fd = d["meta"]["H"], d["meta"]["F"]

    def lin(name, i, o):
        l = torch.nn.Linear(i, o)
        with torch.no_grad():
            l.weight.copy_(d[f"pre_{name}_w"])
            l.bias.copy_(d[f"pre_{name}_b"])
        return l.to(DEV)

    def ln(name, c):
        l = torch.nn.LayerNorm(c)
        with torch.no_grad():
            l.weight.copy_(d[f"pre_{name}_w"])
            l.bias.copy_(d[f"pre_{name}_b"])
        return l.to(DEV)

    m = types.SimpleNamespace()
    m.ln1, m.ln2 = OmniLayerNorm(ln("ln1", H)), OmniLayerNorm(ln("ln2", H))
    m.q, m.k, m.v, m.o = (QuantLinear(lin(n, H, H), WQ, AQ).to(DEV) for n in "qkvo")
    m.fc1, m.fc2 = QuantLinear(lin("fc1", H, Fd), WQ, AQ).to(DEV), QuantLinear(lin("fc2", Fd, H), WQ, AQ).to(DEV)
    m.qkt = QuantMatMul(AQ, AQ, matmul_func=torch.bmm)
    m.pv = QuantMatMul({"n_bits": 16, "metric": "fix0to1"}, AQ, matmul_func=torch.bmm)
    return m


def test_wiring_matches_reference_modules():
    d = golden("reorder")["wiring_small"]
    m = build(d)
    idx = {k: d[k].to(DEV) for k in ("i1", "i2", "i3", "i4", "i5")}
    c1, c2, c5 = (d[k].numpy() for k in ("c1", "c2", "c5"))
    R.R1_reorder(m.ln1, m.q, m.k, m.v, idx["i1"], c1)
    R.R2_reorder(m.q, m.k, m.qkt, idx["i2"], c2)
    R.R3_reorder(m.v, m.pv, m.o, idx["i3"], c2)
    R.R4_reorder(m.ln2, m.fc1, idx["i4"], c1)
    R.R5_reorder(m.fc1, m.fc2, idx["i5"], c5)
    for n in ("ln1", "ln2", "q", "k", "v", "o", "fc1", "fc2"):
        mod = getattr(m, n)
        assert same(mod.weight.cpu(), d[f"post_{n}_w"]), n
        assert same(mod.bias.cpu(), d[f"post_{n}_b"]), n
    assert torch.equal(m.ln1.reorder_index.cpu(), d["ln1_reorder_index"])
    assert torch.equal(m.ln2.reorder_index.cpu(), d["ln2_reorder_index"])
    wired = {"q_act": m.q.act_quantizer, "k_act": m.k.act_quantizer, "v_act": m.v.act_quantizer,
             "o_act": m.o.act_quantizer, "fc1_act": m.fc1.act_quantizer, "fc2_act": m.fc2.act_quantizer,
             "qkt_x1": m.qkt.x1_quantizer, "qkt_x2": m.qkt.x2_quantizer, "pv_x1": m.pv.x1_quantizer,
             "pv_x2": m.pv.x2_quantizer}
    for n, q in wired.items():
        assert (-1 if q.cluster_dim is None else int(q.cluster_dim)) == d[f"wire_{n}_dim"], n
        got = np.zeros(0, dtype=np.int64) if q.cluster_counts is None else np.asarray(q.cluster_counts)
        assert np.array_equal(got, d[f"wire_{n}_counts"].numpy()), n
    # gather + layer norm (layer_norm itself is a torch kernel: equal to rounding)
    y3 = m.ln1(d["x3"].to(DEV))
    assert rel_err(y3, d["ln1_out3"]) <= 2e-6
    y2 = m.ln2(d["x2"].to(DEV))
    assert rel_err(y2, d["ln2_out2"]) <= 2e-6
    # full QuantLinear forward: weight fake-quant (LWC at its initial value) + per-cluster activation
    # quantization + GEMM, fed with the reference's own layer-norm output
    m.q.set_quant_state(True, True)
    out = m.q(d["ln1_out3"].to(DEV))
    assert rel_err(out, d["q_out3"]) <= 1e-5
    m.fc1.set_quant_state(True, True)
    out = m.fc1(d["ln2_out2"].to(DEV))
    assert rel_err(out, d["fc1_out2"]) <= 1e-5
    # the gather is differentiable (inverse permutation)
    xg = d["x3"].to(DEV).requires_grad_(True)
    m.ln1(xg).sum().backward()
    assert xg.grad is not None and torch.isfinite(xg.grad).all()


@pytest.mark.parametrize("name", sorted(k for k in golden("reorder") if k.startswith("h_")))
def test_stat_hooks_bit_exact(name):
    d = golden("reorder")[name]
    m = d["meta"]
    R.oc_maxmin_dict.clear()
    R.ic_maxmin_dict.clear()
    probe = types.SimpleNamespace(name="probe")
    for k, s in enumerate(m["shapes"]):
        t = synth.activation(tuple(s), m["seed"] + k, DT[m["dtype"]]).to(DEV)
        if m["which"] == "out":
            R.layer_omax_hook(probe, (), t)
        else:
            R.layer_i0max_hook(probe, (t,), None)
    store = R.oc_maxmin_dict if m["which"] == "out" else R.ic_maxmin_dict
    assert same(store["probe"][0].cpu(), d["max"]) and same(store["probe"][1].cpu(), d["min"])
    R.oc_maxmin_dict.clear()
    R.ic_maxmin_dict.clear()


@pytest.mark.parametrize("name", ["r_768_c4_h12_list", "r_768_c32_h1"])
def test_tensor_calc_reorder_index_api(name):
    d = golden("reorder")[name]
    m = d["meta"]
    if m["as_list"]:
        idx, counts = R.tensor_calc_reorder_index([d["xmax"].to(DEV), d["kmax"].to(DEV)], [d["xmin"].to(DEV), d["kmin"].to(DEV)],
                                                  m["n_clusters"], m["n_heads"])
    else:
        idx, counts = R.tensor_calc_reorder_index(d["xmax"].to(DEV), d["xmin"].to(DEV), m["n_clusters"], m["n_heads"] or None)
    assert torch.equal(idx.cpu(), d["index"]) and np.array_equal(counts, d["counts"].numpy())


def test_quant_linear_state_machine():
    lin = torch.nn.Linear(64, 32).to(DEV)
    ql = QuantLinear(lin, WQ, {**AQ, "dynamic_method": "per_tensor"})
    ql = ql.to(DEV)
    x = torch.randn(2, 8, 64, device=DEV)
    y0 = ql(x)
    assert torch.equal(y0, torch.nn.functional.linear(x, lin.weight, lin.bias))
    ql.set_quant_state(weight_quant=True, act_quant=False)
    y1 = ql(x)
    assert not torch.equal(y0, y1)
    ql.temp_weight = ql.weight_quantizer(ql.weight)
    ql.temp_bias = ql.bias
    ql.use_temporary_parameter = True
    assert torch.equal(ql(x), y1)
    names = [n for n, _ in ql.named_parameters()]
    assert "weight_quantizer.upbound_factor" in names and "weight_quantizer.lowbound_factor" in names
    assert ql.weight_quantizer.upbound_factor.shape == (32, 1)
    ql.weight_quantizer.register_scales_and_zeros()
    assert "scales" in dict(ql.weight_quantizer.named_buffers())


def test_fake_quant_reads_the_layout_from_the_scale_shape():
    """A [rows, 1] scale on a SQUARE weight (what per_token_dynamic_calibration leaves in ``scale``) is a per-row scale,
    not a per-column one; per-channel and scalar scales keep working; anything else is refused."""
    from miq.quantizer import UniformAffineQuantizer
    from miq._lib import MiqError
    w = synth.weight(96, 96, 21).to(DEV)
    q = UniformAffineQuantizer(n_bits=4, per_channel_axes=[0], dynamic=True, dynamic_method="per_channel", lwc=False,
                               shape=(96, 96)).to(DEV)
    out = q(w)
    assert q.scale.shape == (96, 1)
    again = q.fake_quant(w, q.scale, q.round_zero_point)
    assert torch.equal(again, out)
    ref = ((w / q.scale).round() + q.round_zero_point).clamp(0, 15).sub(q.round_zero_point).mul(q.scale)
    assert torch.equal(again, ref)
    col_scale = q.scale.reshape(1, 96).contiguous()
    col = q.fake_quant(w, col_scale, q.round_zero_point.reshape(1, 96).contiguous())
    ref_col = ((w / col_scale).round() + q.round_zero_point.reshape(1, 96)).clamp(0, 15).sub(q.round_zero_point.reshape(1, 96)).mul(col_scale)
    assert torch.equal(col, ref_col) and not torch.equal(col, again)
    one = q.fake_quant(w, q.scale[:1, 0].reshape(()), q.round_zero_point[:1, 0].reshape(()))
    assert one.shape == w.shape
    with pytest.raises(MiqError):
        q.fake_quant(w, q.scale[:48], q.round_zero_point[:48])

