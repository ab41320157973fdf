"""Host-side logic that needs no GPU: constructor/attribute surface of the drop-in modules, the
quantizer keyword dictionaries, checkpoint format helpers, and the loud failure for CPU tensors."""
import numpy as np
import pytest
import torch

from miq import functional as F
from miq._lib import MiqError
from miq.blocks import QuantOPTBlock, quant_args
from miq.calibrate import load_omni_parameters, save_omni_parameters
from miq.int_linear import QuantLinear
from miq.int_matmul import QuantMatMul
from miq.quantizer import UniformAffineQuantizer
from miq.quantizer_exp import ExpAffineQuantizer as Exp1Q
from miq.quantizer_exp_v2 import ExpAffineQuantizer as Exp2Q
from miq.quantizer_mix import MixAffineQuantizer
from miq.quantizer_mix_v2 import Mix2AffineQuantizer


def test_uniform_surface():
    q = UniformAffineQuantizer(n_bits=3, symmetric=False, per_channel_axes=[0], metric="minmax", dynamic=True,
                               dynamic_method="per_channel", group_size=None, shape=(768, 768), lwc=True, target_bit=3)
    assert (q.qmin, q.qmax, q.mode, q.enable, q.deficiency) == (0, 7, "calibration", True, 0)
    assert q.upbound_factor.shape == (768, 1) and float(q.upbound_factor[0]) == 4.0
    assert [n for n, _ in q.named_parameters()] == ["upbound_factor", "lowbound_factor"]
    assert q.scale is None and q.cached_xmax is None and q.cluster_counts is None
    q.change_n_bits(4)
    assert q.qmax == 15
    q.set_eval_mode()
    assert q.mode == "eval"
    g = UniformAffineQuantizer(n_bits=4, symmetric=True, dynamic=True, dynamic_method="per_channel", group_size=128,
                               shape=(16, 328), lwc=True)
    assert g.upbound_factor.shape == (16 * 3, 1) and g.deficiency == 128 - 328 % 128
    with pytest.raises(AssertionError):
        UniformAffineQuantizer(n_bits=1)
    with pytest.raises(AssertionError):  # zero padding needs a symmetric grid
        UniformAffineQuantizer(n_bits=4, symmetric=False, group_size=128, shape=(16, 328), lwc=True)


def test_passthrough_paths_need_no_gpu():
    x = torch.randn(4, 8)
    assert UniformAffineQuantizer(n_bits=16)(x) is x
    q = UniformAffineQuantizer(n_bits=8)
    q.enable = False
    assert q(x) is x


def test_cpu_tensors_fail_loudly():
    x = torch.randn(4, 8)
    with pytest.raises(MiqError):
        UniformAffineQuantizer(n_bits=4, dynamic=True, dynamic_method="per_channel")(x)
    with pytest.raises(MiqError):
        UniformAffineQuantizer(n_bits=4, dynamic=False, dynamic_method="per_tensor")(x)
    with pytest.raises(MiqError):
        F.gather_lastdim(x, torch.arange(8, dtype=torch.int32))


def test_other_quantizers_surface():
    e2 = Exp2Q(n_bits=3, symmetric=True, shape=(8, 8), lwc=True)
    assert (e2.qmin, e2.qmax) == (-4, 4)
    assert (Exp2Q(n_bits=4).qmax, Exp2Q(n_bits=2).qmax) == (64, 1)
    e1 = Exp1Q(n_bits=3)
    assert (e1.qmin, e1.qmax) == (-3, 0)
    with pytest.raises(AssertionError):
        Exp1Q(n_bits=5)
    m = MixAffineQuantizer(target_bit=3, n_bits=[], shape=(16, 32), lwc=True)
    assert isinstance(m.n_bits, torch.nn.Parameter) and m.n_bits.shape == (16, 1)
    assert float(m.n_bits.min()) >= 3.5 and float(m.n_bits.max()) <= 3.75
    m.set_hard()
    assert m.hard_assign
    m2 = Mix2AffineQuantizer(mix2_nbits=torch.tensor([2.0, 3.0, 4.0]), shape=(3, 8))
    assert m2.n_bits.shape == (3, 1) and torch.equal(m2.q_max.flatten(), torch.tensor([3.0, 7.0, 15.0]))


def test_wrappers_and_block_surface():
    a = quant_args(wbits=4, abits=4)
    assert a.weight_quant_params["dynamic"] and a.act_quant_params["dynamic_method"] == "per_cluster"
    assert a.p_quant_params == {"n_bits": 16, "metric": "fix0to1"}
    lin = torch.nn.Linear(8, 4)
    ql = QuantLinear(lin, a.weight_quant_params, a.act_quant_params)
    assert not isinstance(ql.weight, torch.nn.Parameter) and ql.weight.data_ptr() == lin.weight.data_ptr()
    ql.set_ic_cluster_counts(np.array([4, 4]), a_dim=2)
    assert ql.act_quantizer.cluster_dim == 2 and ql.weight_quantizer.cluster_dim == 1
    mm = QuantMatMul(a.q_quant_params, a.k_quant_params)
    mm.set_ic_cluster_counts(np.array([2, 2]), cluster_x1=False)
    assert mm.x1_quantizer.cluster_counts is None and mm.x2_quantizer.cluster_dim == 2
    H, Fd = 16, 32
    g = torch.Generator().manual_seed(0)
    w = {n: torch.randn(H, H, generator=g) for n in "qkvo"}
    w.update({f"{n}_b": torch.zeros(H) for n in "qkvo"})
    w.update(fc1=torch.randn(Fd, H, generator=g), fc1_b=torch.zeros(Fd), fc2=torch.randn(H, Fd, generator=g), fc2_b=torch.zeros(H),
             ln1_w=torch.ones(H), ln1_b=torch.zeros(H), ln2_w=torch.ones(H), ln2_b=torch.zeros(H))
    blk = QuantOPTBlock(w, 4, a)
    names = [n for n, _ in blk.named_parameters()]
    assert "self_attn.q_proj.weight_quantizer.upbound_factor" in names and "fc2.weight_quantizer.lowbound_factor" in names
    assert len(list(blk.lwc_parameters())) == 12 and len(list(blk.let_parameters())) == 0
    blk.register_parameter("qkv_smooth_scale", torch.nn.Parameter(torch.ones(H)))
    assert len(list(blk.let_parameters())) == 1 and len(list(blk.omni_parameters())) == 13
    assert "qkv_smooth_scale" in blk.omni_state_dict()
    # quantization off: the block is a plain fp block and runs on CPU
    y = blk(torch.randn(1, 5, H, generator=g))[0]
    assert y.shape == (1, 5, H)


def test_omni_parameters_checkpoint_roundtrip(tmp_path):
    state = {0: {"qkv_smooth_scale": torch.ones(4).half(), "fc1.weight_quantizer.upbound_factor": torch.full((4, 1), 4.0).half()}}
    save_omni_parameters(state, str(tmp_path))
    back = load_omni_parameters(str(tmp_path / "omni_parameters.pth"))
    assert list(back[0].keys()) == list(state[0].keys()) and torch.equal(back[0]["qkv_smooth_scale"], state[0]["qkv_smooth_scale"])


def test_reference_checkpoint_keys_match_block_names():
    """The parameter names of QuantOPTBlock are the keys of the reference's omni_parameters.pth
    (19 tensors per layer for OPT: 12 LWC factors + 7 LET vectors)."""
    expected = {"qkt_smooth_scale", "qkv_smooth_shift", "qkv_smooth_scale", "out_smooth_shift", "out_smooth_scale",
                "fc1_smooth_shift", "fc1_smooth_scale"}
    for lin in ("self_attn.k_proj", "self_attn.v_proj", "self_attn.q_proj", "self_attn.out_proj", "fc1", "fc2"):
        expected.add(f"{lin}.weight_quantizer.upbound_factor")
        expected.add(f"{lin}.weight_quantizer.lowbound_factor")
    assert len(expected) == 19
    from helpers import golden
    assert set(golden("block")["opt_tiny_w4a4_reorder"]["omni_keys"]) == expected


def test_reference_shipped_checkpoint_loads_into_block():
    """tests/golden/ref_omni_parameters_opt125m_w3a16.pth is the omni_parameters.pth the reference
    ships in log/opt-125m-w3a16/ (data, 12 layers x 19 fp16 tensors).  It must load (safe loader) into
    an OPT-125m-shaped block exactly like the reference's --resume does (load_state_dict, strict=False)."""
    import os
    from helpers import GOLDEN
    ck = load_omni_parameters(os.path.join(GOLDEN, "ref_omni_parameters_opt125m_w3a16.pth"))
    assert sorted(ck.keys()) == list(range(12))
    H, Fd = 768, 3072
    w = {n: torch.zeros(H, H) for n in "qkvo"}
    w.update({f"{n}_b": torch.zeros(H) for n in "qkvo"})
    w.update(fc1=torch.zeros(Fd, H), fc1_b=torch.zeros(Fd), fc2=torch.zeros(H, Fd), fc2_b=torch.zeros(H),
             ln1_w=torch.ones(H), ln1_b=torch.zeros(H), ln2_w=torch.ones(H), ln2_b=torch.zeros(H))
    blk = QuantOPTBlock(w, 12, quant_args(wbits=3, abits=16, w_symmetric=True, weight_exp_quant=True))
    for tag in ("qkv", "out", "fc1"):
        blk.register_parameter(f"{tag}_smooth_shift", torch.nn.Parameter(torch.zeros(H)))
        blk.register_parameter(f"{tag}_smooth_scale", torch.nn.Parameter(torch.ones(H)))
    blk.register_parameter("qkt_smooth_scale", torch.nn.Parameter(torch.ones(H)))
    missing, unexpected = blk.load_state_dict(ck[3], strict=False)
    assert not unexpected
    assert all(("smooth" not in m and "bound_factor" not in m) for m in missing)
    assert torch.equal(blk.fc1.weight_quantizer.upbound_factor.detach().half(), ck[3]["fc1.weight_quantizer.upbound_factor"])
    assert set(blk.omni_state_dict().keys()) == set(ck[3].keys())


def test_reference_act_scale_shift_files_feed_let_init():
    """The act_scales / act_shifts files the reference ships for OPT-125m (73 entries keyed by module
    name, float32 [in_features]) are consumed as they are by the LET initialisation."""
    import os
    from miq.blocks import QuantOPTBlock, quant_args
    from miq.calibrate import CalibConfig, init_let
    here = os.path.join(os.path.dirname(__file__), "golden")
    scales = torch.load(os.path.join(here, "ref_act_scales_opt125m.pt"), weights_only=True)
    shifts = torch.load(os.path.join(here, "ref_act_shifts_opt125m.pt"), weights_only=True)
    assert len(scales) == len(shifts) == 73 and set(scales) == set(shifts)
    H, Fd = 768, 3072
    assert scales["model.decoder.layers.0.fc2"].shape == (Fd,) and scales["model.decoder.layers.11.fc1"].dtype == torch.float32
    g = torch.Generator().manual_seed(0)
    w = {}
    for n, (o, i) in {"q": (H, H), "k": (H, H), "v": (H, H), "o": (H, H), "fc1": (Fd, H), "fc2": (H, Fd)}.items():
        w[n] = torch.randn(o, i, generator=g) * 0.02
        w[f"{n}_b"] = torch.zeros(o)
    for n in ("ln1", "ln2"):
        w[f"{n}_w"], w[f"{n}_b"] = torch.ones(H), torch.zeros(H)
    q = QuantOPTBlock(w, 12, quant_args(let=True))
    init_let(q, scales, shifts, "model.decoder.layers.3", CalibConfig(), None, torch.float32, torch.device("cpu"))
    names = {n for n, _p in q.named_parameters() if "smooth" in n}
    assert names == {"qkt_smooth_scale", "qkv_smooth_scale", "qkv_smooth_shift", "out_smooth_scale", "out_smooth_shift",
                     "fc1_smooth_scale", "fc1_smooth_shift"}
    assert torch.equal(q.fc1_smooth_shift.detach(), shifts["model.decoder.layers.3.fc1"])
    assert bool((q.qkv_smooth_scale > 0).all()) and bool(torch.isfinite(q.out_smooth_scale).all())


def test_grad_norm_matches_the_reference_formula():
    """Norm of per-parameter norms (reference utils.py:11-24), also when some parameters have no gradient."""
    import torch
    from miq.calibrate import grad_norm

    g = torch.Generator().manual_seed(3)
    ps = [torch.nn.Parameter(torch.randn(n, generator=g)) for n in (7, 64, 1, 300)]
    for p in ps[:3]:
        p.grad = torch.randn(p.shape, generator=g)
    ref = torch.norm(torch.stack([torch.norm(p.grad.detach(), 2.0) for p in ps if p.grad is not None]), 2.0)
    assert torch.allclose(grad_norm(ps), ref, rtol=1e-6)
    assert float(grad_norm([ps[3]])) == 0.0
