"""Llama block calibration on the GPU against a fixture captured from the reference's own
QuantLlamaDecoderLayer + the llama branch of omniquant() (tests/golden/block_llama.pt): tiny
Llama-shaped block (hidden 64, SwiGLU 160, 4 heads, rotary, RMSNorm, 32 tokens, 4 samples, 2 epochs),
W4A4 with the reference's default static per-tensor EMA activation quantizers, LET (zero shifts) +
LWC, fp32 path.

Tolerances as for the OPT block: the quantizer kernels are bit-exact, the GEMMs / softmax / norms
around them run on another device than the fixture, and 8 optimizer steps amplify that rounding."""
import pytest
import torch

from helpers import golden, rel_err
from miq.blocks import QuantLlamaBlock, quant_args
from miq.calibrate import CalibConfig, calibrate_layer, finalize_layer, init_let

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PREFIX = "model.layers.0"
LET_SITES = ("self_attn.q_proj", "self_attn.o_proj", "mlp.up_proj")


def build(d):
    m = d["meta"]
    w = {n: d[f"w_{n}"].to(DEV) for n in ("q", "k", "v", "o", "gate", "up", "down", "ln1_w", "ln2_w")}
    args = quant_args(wbits=m["wbits"], abits=m["abits"], lwc=True, a_dynamic_method="per_tensor", metric="ema_minmax",
                      let=m.get("let", True))
    return QuantLlamaBlock(w, m["heads"], args, num_kv_heads=m.get("kv_heads", m["heads"]), eps=m["eps"],
                           rope_theta=m["theta"]).to(DEV)


def config(m, **kw):
    # autocast_passes=False: the fixtures come from a CPU run of the reference, where the full-precision and the
    # propagation pass are fp32 (the reference's CUDA autocast of those passes has no CPU counterpart)
    return CalibConfig(epochs=m["epochs"], nsamples=m["N"], batch_size=1, let=m.get("let", True), lwc=True, deactive_amp=True,
                       reorder="", autocast_passes=False, **kw)


def test_tiny_llama_block_matches_reference_run():
    d = golden("block_llama")["llama_tiny_w4a4_pertensor"]
    m = d["meta"]
    q = build(d)
    act_scales = {f"{PREFIX}.{n}": d[f"act_scale_{n}"] for n in LET_SITES}
    quant_inps = d["inps"].clone().to(DEV)
    fp_inps = d["inps"].clone().to(DEV)
    res = calibrate_layer(q, quant_inps, fp_inps, d["mask"].to(DEV), config(m), act_scales, {}, PREFIX, is_llama=True)
    assert rel_err(fp_inps, d["fp_out"]) <= 1e-5
    losses, norms = torch.tensor(res["losses"]), torch.tensor(res["norms"])
    assert torch.allclose(losses, d["losses"], rtol=2e-2), (losses, d["losses"])
    assert torch.allclose(norms, d["norms"], rtol=5e-2), (norms, d["norms"])
    state = res["omni_state"]
    assert list(state.keys()) == d["omni_keys"]
    for k in d["omni_keys"]:
        ref, got = d[f"omni_fp32_{k}"], state[k].float().cpu()
        assert got.shape == ref.shape, k
        # AdamW moves every element by about +-lr per step: an element whose gradient sign flips once (a
        # quantization code flipped by GEMM / LayerNorm rounding on another device) ends up to 2*lr away, so
        # the bulk must agree tightly and single elements may differ by a few learning-rate steps
        err = (got - ref).abs()
        bound = 2e-3 + 1e-3 * ref.abs().max().item()
        assert (err <= bound).float().mean().item() >= 0.98, k
        assert err.max().item() <= 4e-2, k
    qo = quant_inps.cpu()
    assert ((qo - d["quant_out"]).abs().mean() / d["quant_out"].abs().mean()).item() <= 5e-2


@pytest.mark.parametrize("case,abits", [("llama_tiny_gqa_w4a4_nolet", 4), ("llama_tiny_gqa_w4a8_nolet", 8)])
def test_tiny_gqa_llama_block_matches_reference_run(case, abits):
    """Grouped-query attention (2 key / value heads for 4 query heads: the Llama-2-70B layout, repeat_kv of
    models/int_llama_layer.py:130-137) without LET, against the reference's QuantLlamaDecoderLayer driven by the llama
    branch of its loop (tests/golden/block_llama_gqa.pt).  Only the clipping factors train.  The second case is the
    arithmetic of the Llama-2-70B W4A8 configuration: grouped-query attention AND 8-bit activations (fp32 path)."""
    d = golden("block_llama_gqa")[case]
    m = d["meta"]
    assert m["kv_heads"] == 2 and m["heads"] == 4 and not m["let"] and m["abits"] == abits
    q = build(d)
    assert q.self_attn.k_proj.weight.shape == (32, 64)
    quant_inps = d["inps"].clone().to(DEV)
    fp_inps = d["inps"].clone().to(DEV)
    res = calibrate_layer(q, quant_inps, fp_inps, d["mask"].to(DEV), config(m), {}, {}, PREFIX, is_llama=True)
    assert rel_err(fp_inps, d["fp_out"]) <= 1e-5
    losses, norms = torch.tensor(res["losses"]), torch.tensor(res["norms"])
    assert torch.allclose(losses, d["losses"], rtol=2e-2), (losses, d["losses"])
    assert torch.allclose(norms, d["norms"], rtol=5e-2), (norms, d["norms"])
    state = res["omni_state"]
    assert list(state.keys()) == d["omni_keys"]
    for k in d["omni_keys"]:
        ref, got = d[f"omni_fp32_{k}"], state[k].float().cpu()
        assert got.shape == ref.shape, k
        err = (got - ref).abs()
        bound = 2e-3 + 1e-3 * ref.abs().max().item()
        assert (err <= bound).float().mean().item() >= 0.98, k
        assert err.max().item() <= 4e-2, k
    qo = quant_inps.cpu()
    assert ((qo - d["quant_out"]).abs().mean() / d["quant_out"].abs().mean()).item() <= 5e-2
    for name, mod in q.named_modules():
        if f"actscale_{name}" in d:
            assert rel_err(mod.act_quantizer.scale.float().cpu().reshape(-1), d[f"actscale_{name}"].reshape(-1)) <= 2e-2, name


def test_reference_autocast_of_the_fp_and_propagation_passes():
    """With deactive_amp the reference still runs the full-precision target pass and the final propagation pass under
    fp16 autocast (quantize/omniquant.py:402, :583).  The default configuration does the same: the targets then carry
    fp16 GEMM rounding (close to, but not equal to, the fp32 pass) and the run stays finite."""
    d = golden("block_llama")["llama_tiny_w4a4_pertensor"]
    m = d["meta"]
    act_scales = {f"{PREFIX}.{n}": d[f"act_scale_{n}"] for n in LET_SITES}
    outs = {}
    for passes in (False, True):
        q = build(d).half()  # the block as the reference holds it at that point: fp16 weights
        cfg = config(m)
        cfg.autocast_passes = passes
        quant_inps, fp_inps = d["inps"].clone().to(DEV), d["inps"].clone().to(DEV)
        res = calibrate_layer(q, quant_inps, fp_inps, d["mask"].to(DEV), cfg, act_scales, {}, PREFIX, is_llama=True)
        assert fp_inps.dtype == torch.float32 and torch.isfinite(fp_inps).all() and torch.isfinite(quant_inps).all()
        assert all(map(lambda v: v == v, res["losses"]))
        outs[passes] = (fp_inps.clone(), quant_inps.clone())
    err = rel_err(outs[True][0], outs[False][0])
    assert 0 < err <= 5e-3, err  # fp16 GEMMs: different from the fp32 pass, but only by their rounding


@pytest.mark.parametrize("kind", ["llama", "opt"])
def test_shared_and_batched_activation_quantizers_equal_one_call_each(kind):
    """q / k / v (and gate / up) read one tensor through three identically configured quantizers: evaluating the
    quantizer once, and quantizing the three projection outputs in one launch pair, must give the same loss to the last
    bit and the same parameter gradients up to the order in which three gradient contributions are summed."""
    import miq.blocks as B
    if kind == "llama":
        d = golden("block_llama")["llama_tiny_w4a4_pertensor"]
        builder, prefix, sites, is_llama, shifts = build, PREFIX, LET_SITES, True, {}
    else:
        import test_gpu_block_calibration as opt
        d = golden("block")["opt_tiny_w4a4_reorder"]
        builder, prefix, is_llama = opt.build, "model.decoder.layers.0", False
        sites = ("self_attn.q_proj", "self_attn.out_proj", "fc1")
        shifts = {f"{prefix}.{n}": d[f"act_shift_{n}"] for n in sites}
    m = d["meta"]
    cfg = config(m)
    act_scales = {f"{prefix}.{n}": d[f"act_scale_{n}"] for n in sites}
    x, y = d["inps"][:2].to(DEV), d["fp_out"][:2].to(DEV)
    mask = d["mask"].to(DEV).repeat(2, 1, 1, 1)
    res = {}
    try:
        for share in (True, False):
            B.SHARE_ACT_QUANT = share
            q = builder(d)
            q.set_quant_state(weight_quant=False, act_quant=True)
            q.let = True
            init_let(q, act_scales, shifts, prefix, cfg, None, torch.float32, torch.device(DEV), is_llama=is_llama)
            with torch.no_grad():
                q.float()
            losses = []
            for _step in range(2):  # the second step runs with freed statistics again, like the training loop
                q.zero_grad()
                q.smooth_and_quant_temporary()
                loss = torch.nn.functional.mse_loss(q(x, attention_mask=mask)[0], y)
                loss.backward()
                losses.append(loss.item())
                for aq in q.act_quantizers():
                    aq.free()
            res[share] = (losses, {n: p.grad.clone() for n, p in q.named_parameters() if p.grad is not None},
                          [aq.scale.clone() for aq in q.act_quantizers() if aq.scale is not None])
    finally:
        B.SHARE_ACT_QUANT = True
    assert res[True][0] == res[False][0]
    assert len(res[True][2]) == len(res[False][2]) >= 6
    for a, b in zip(res[True][2], res[False][2]):
        assert torch.equal(a, b)
    assert res[True][1].keys() == res[False][1].keys()
    for n, g in res[True][1].items():
        assert rel_err(g, res[False][1][n]) <= 2e-5, n


def test_llama_propagation_with_reference_parameters():
    """The reference's learned parameters loaded, then only the tail of the layer procedure: quantized
    weights, accumulated per-tensor activation scales and propagated activations must agree closely."""
    d = golden("block_llama")["llama_tiny_w4a4_pertensor"]
    m = d["meta"]
    q = build(d)
    cfg = config(m)
    act_scales = {f"{PREFIX}.{n}": d[f"act_scale_{n}"] for n in LET_SITES}
    q.set_quant_state(weight_quant=False, act_quant=True)
    q.let = True
    init_let(q, act_scales, {}, PREFIX, cfg, None, torch.float32, torch.device(DEV), is_llama=True)
    state = {k: d[f"omni_fp32_{k}"].to(DEV) for k in d["omni_keys"]}
    _missing, unexpected = q.load_state_dict(state, strict=False)
    assert not unexpected
    quant_inps = d["inps"].clone().to(DEV)
    finalize_layer(q, quant_inps, d["mask"].to(DEV), cfg)
    for name, mod in q.named_modules():
        if f"wq_{name}" in d:
            w, r = mod.weight.float().cpu(), d[f"wq_{name}"]
            assert ((w - r.half().float()).abs() > 1e-6).float().mean().item() <= 2e-3, name
        if f"actscale_{name}" in d:
            assert rel_err(mod.act_quantizer.scale.float().cpu().reshape(-1), d[f"actscale_{name}"].reshape(-1)) <= 2e-3, name
    qo = quant_inps.cpu()
    assert ((qo - d["quant_out"]).abs().mean() / d["quant_out"].abs().mean()).item() <= 2e-3


@pytest.mark.parametrize("kind", ["llama", "opt"])
def test_fused_let_step_equals_unfused_step(kind):
    """One training step with the LET-fused weight-quantizer kernels against the same step through the
    reference's op chain (smooth_*_temporary, then the stand-alone quantizer kernels): the quantized
    weights are bit-identical, the loss agrees to fp32 rounding of the bias terms, and every LET / LWC
    gradient agrees to 1e-4 of its norm."""
    from miq.int_linear import QuantLinear
    if kind == "llama":
        d = golden("block_llama")["llama_tiny_w4a4_pertensor"]
        builder, prefix, sites, is_llama = build, PREFIX, LET_SITES, True
        shifts = {}
    else:
        import test_gpu_block_calibration as opt
        d = golden("block")["opt_tiny_w4a4_reorder"]
        builder, prefix, is_llama = opt.build, "model.decoder.layers.0", False
        sites = ("self_attn.q_proj", "self_attn.out_proj", "fc1")
        shifts = {f"{prefix}.{n}": d[f"act_shift_{n}"] for n in sites}
    m = d["meta"]
    cfg = config(m)
    act_scales = {f"{prefix}.{n}": d[f"act_scale_{n}"] for n in sites}
    x = d["inps"][:1].to(DEV)
    mask = d["mask"].to(DEV)
    results = []
    for fused in (True, False):
        q = builder(d)
        q.fused_let = fused
        q.set_quant_state(weight_quant=False, act_quant=True)
        q.let = True
        init_let(q, act_scales, shifts, prefix, cfg, None, torch.float32, torch.device(DEV), is_llama=is_llama)
        with torch.no_grad():
            q.float()
            for n, p in q.named_parameters():  # move the parameters off their initial values
                if "smooth_shift" in n:
                    p.add_(0.05 * torch.sin(torch.arange(p.numel(), device=DEV, dtype=torch.float32)))
                if "qkt_smooth_scale" in n:
                    p.mul_(1.0 + 0.1 * torch.cos(torch.arange(p.numel(), device=DEV, dtype=torch.float32)))
        assert q._let_fusable() == fused
        q.smooth_and_quant_temporary()
        out = q(x, attention_mask=mask)[0]
        loss = (out ** 2).mean()
        names = [n for n, _p in q.named_parameters() if "smooth" in n or "bound_factor" in n]
        grads = torch.autograd.grad(loss, [dict(q.named_parameters())[n] for n in names])
        wq = {n: mod.temp_weight.detach().clone() for n, mod in q.named_modules() if isinstance(mod, QuantLinear)}
        results.append((float(loss.detach()), dict(zip(names, grads)), wq))
    (l1, g1, w1), (l0, g0, w0) = results
    for n in w0:
        assert torch.equal(w1[n], w0[n]), n
    assert abs(l1 - l0) <= 1e-5 * abs(l0)
    for n in g0:
        assert rel_err(g1[n].cpu(), g0[n].cpu()) <= 1e-4, (n, rel_err(g1[n].cpu(), g0[n].cpu()))


def test_fused_adamw_option_trains_close_to_the_default():
    """CalibConfig.fused_adamw only changes the optimizer's arithmetic in the last bits: the first loss is identical
    and the run stays within the block's usual rounding amplification of the default run."""
    import dataclasses
    d = golden("block_llama")["llama_tiny_w4a4_pertensor"]
    m = d["meta"]
    act_scales = {f"{PREFIX}.{n}": d[f"act_scale_{n}"] for n in LET_SITES}
    runs = {}
    for fused in (False, True):
        q = build(d)
        cfg = dataclasses.replace(config(m), fused_adamw=fused)
        runs[fused] = calibrate_layer(q, d["inps"].clone().to(DEV), d["inps"].clone().to(DEV), d["mask"].to(DEV), cfg, act_scales, {},
                                      PREFIX, is_llama=True)
    a, b = torch.tensor(runs[False]["losses"]), torch.tensor(runs[True]["losses"])
    assert a[0] == b[0]
    assert torch.allclose(a, b, rtol=3e-2), (a, b)


def test_aug_loss_and_resume():
    """aug_loss adds the MSE against the fp block applied to the quantized model's inputs: for a first
    layer both inputs coincide, so the first loss doubles exactly.  resume_state reloads learned
    parameters before training (reference --resume)."""
    import dataclasses
    d = golden("block_llama")["llama_tiny_w4a4_pertensor"]
    m = d["meta"]
    act_scales = {f"{PREFIX}.{n}": d[f"act_scale_{n}"] for n in LET_SITES}
    runs = {}
    for aug in (False, True):
        q = build(d)
        cfg = dataclasses.replace(config(m), aug_loss=aug, epochs=1)
        x = d["inps"].clone().to(DEV)
        runs[aug] = calibrate_layer(q, x, d["inps"].clone().to(DEV), d["mask"].to(DEV), cfg, act_scales, {}, PREFIX, is_llama=True)
    assert abs(runs[True]["losses"][0] - 2 * runs[False]["losses"][0]) <= 1e-6 * runs[False]["losses"][0]
    # resume: zero epochs of training starting from the first run's parameters reproduces them
    q = build(d)
    cfg = dataclasses.replace(config(m), epochs=1, let_lr=0.0, lwc_lr=0.0)
    state = {k: v.float() for k, v in runs[False]["omni_state"].items()}
    res = calibrate_layer(q, d["inps"].clone().to(DEV), d["inps"].clone().to(DEV), d["mask"].to(DEV), cfg, act_scales, {},
                          PREFIX, is_llama=True, resume_state=state)
    for k, v in res["omni_state"].items():
        assert torch.equal(v.cpu(), runs[False]["omni_state"][k].cpu()), k


def test_layer_loop_two_blocks_saves_and_resumes(tmp_path):
    """calibrate_blocks over two blocks: streams are chained, omni_parameters.pth is written per layer in
    the reference's format and a resumed run starts from the stored parameters."""
    import dataclasses
    from miq.calibrate import calibrate_blocks, load_omni_parameters
    d = golden("block_llama")["llama_tiny_w4a4_pertensor"]
    m = d["meta"]
    cfg = dataclasses.replace(config(m), epochs=1)
    scales = {}
    for i in range(2):
        for n in LET_SITES:
            scales[f"model.layers.{i}.{n}"] = d[f"act_scale_{n}"]
    blocks = [build(d), build(d)]
    x0 = d["inps"].clone().to(DEV)
    params, results, out = calibrate_blocks(blocks, x0, d["mask"].to(DEV), cfg, scales, {}, "model.layers", is_llama=True,
                                            output_dir=str(tmp_path))
    assert sorted(params.keys()) == [0, 1] and len(results) == 2 and out.shape == x0.shape
    assert not torch.equal(out.cpu(), d["inps"])  # the stream moved through both blocks
    stored = load_omni_parameters(str(tmp_path / "omni_parameters.pth"))
    assert sorted(stored.keys()) == [0, 1] and list(stored[1].keys()) == d["omni_keys"]
    for k in d["omni_keys"]:
        assert stored[0][k].dtype == torch.float16 and torch.equal(stored[0][k].cpu(), params[0][k].cpu())
    # resume with zero learning rates: layer 0 reproduces the stored parameters
    cfg0 = dataclasses.replace(cfg, let_lr=0.0, lwc_lr=0.0)
    p2, _r, _o = calibrate_blocks([build(d)], d["inps"].clone().to(DEV), d["mask"].to(DEV), cfg0, scales, {}, "model.layers",
                                  is_llama=True, resume={0: {k: v.float() for k, v in stored[0].items()}})
    for k in d["omni_keys"]:
        assert torch.equal(p2[0][k].cpu(), stored[0][k].cpu()), k


@pytest.mark.parametrize("let", [True, False])
def test_half_weight_path_under_autocast_equals_quantize_then_cast(let):
    """Under fp16 autocast plain fp32 weights are quantized straight into the fp16 tensor the linear uses: the
    temporary weights equal the cast of the fp32 quantized weights bit for bit, so the block output is unchanged,
    and the LWC gradients agree with the cast path to fp16 rounding of the weight gradient."""
    from miq.blocks import _BlockBase
    from miq.int_linear import QuantLinear
    d = golden("block_llama")["llama_tiny_w4a4_pertensor"]
    m = d["meta"]
    cfg = config(m)
    act_scales = {f"{PREFIX}.{n}": d[f"act_scale_{n}"] for n in LET_SITES}
    x = d["inps"][:1].to(DEV).half()
    mask = d["mask"].to(DEV).half()
    results = []
    try:
        for half in (True, False):
            _BlockBase.half_weights = half
            q = build(d)
            q.set_quant_state(weight_quant=False, act_quant=True)
            q.let = let
            if let:
                init_let(q, act_scales, {}, PREFIX, cfg, None, torch.float32, torch.device(DEV), is_llama=True)
            q.float()
            with torch.autocast("cuda", dtype=torch.float16):
                q.smooth_and_quant_temporary()
                out = q(x, attention_mask=mask)[0]
                loss = (out.float() ** 2).mean()
            lin = {n: mod for n, mod in q.named_modules() if isinstance(mod, QuantLinear)}
            plain = [n for n in lin if not let or n.endswith("down_proj")]
            for n in plain:
                assert lin[n].temp_weight.dtype == (torch.float16 if half else torch.float32), n
            names = [n for n, _p in q.named_parameters() if "bound_factor" in n]
            grads = torch.autograd.grad(loss, [dict(q.named_parameters())[n] for n in names])
            results.append((out.detach(), {n: mod.temp_weight.detach().half() for n, mod in lin.items()}, dict(zip(names, grads))))
    finally:
        _BlockBase.half_weights = True
    (o1, w1, g1), (o0, w0, g0) = results
    for n in w0:
        assert torch.equal(w1[n], w0[n]), n
    assert torch.equal(o1, o0)
    for n in g0:
        assert rel_err(g1[n].cpu(), g0[n].cpu()) <= 2e-3, (n, rel_err(g1[n].cpu(), g0[n].cpu()))


@pytest.mark.parametrize("kind", ["llama", "opt_reordered"])
def test_norm_kernels_that_gather_statistics_equal_the_separate_kernels(kind):
    """Layer norm -> shared activation quantizer with the quantizer's statistics folded into the norm kernel (RMSNorm for
    the Llama block with per-tensor statistics; gather + LayerNorm for the reordered OPT block with k-means clusters of
    unequal sizes) against the norm followed by the statistics kernel: same loss to the last bit, same activation
    scales, parameter gradients up to summation order."""
    import miq.quantizer as Q
    from miq import functional as F
    from miq import reorder_utils as R
    if kind == "llama":
        d = golden("block_llama")["llama_tiny_w4a4_pertensor"]
        builder, prefix, sites, is_llama, shifts = build, PREFIX, LET_SITES, True, {}
    else:
        import test_gpu_block_calibration as opt
        from miq.calibrate import reorder_block
        d = golden("block")["opt_tiny_w4a4_reorder"]
        prefix, is_llama = "model.decoder.layers.0", False
        sites = ("self_attn.q_proj", "self_attn.out_proj", "fc1")
        shifts = {f"{prefix}.{n}": d[f"act_shift_{n}"] for n in sites}

        def builder(dd):
            q = opt.build(dd)
            mm = dd["meta"]
            R.oc_maxmin_dict.clear()
            R.ic_maxmin_dict.clear()
            for name in ("self_attn_layer_norm", "self_attn.q_proj", "self_attn.k_proj", "final_layer_norm"):
                R.oc_maxmin_dict[name] = (dd[f"stat_oc_{name}_max"].to(DEV), dd[f"stat_oc_{name}_min"].to(DEV))
            for name in ("self_attn.out_proj", "fc2"):
                R.ic_maxmin_dict[name] = (dd[f"stat_ic_{name}_max"].to(DEV), dd[f"stat_ic_{name}_min"].to(DEV))
            reorder_block(q, mm["H"], mm["F"], mm["heads"], {f"R{k}": True for k in range(1, 6)}, mm["clusters"], DEV)
            R.oc_maxmin_dict.clear()
            R.ic_maxmin_dict.clear()
            return q
    m = d["meta"]
    cfg = config(m)
    act_scales = {f"{prefix}.{n}": d[f"act_scale_{n}"] for n in sites}
    x, y = d["inps"][:2].to(DEV), d["fp_out"][:2].to(DEV)
    mask = d["mask"].to(DEV).repeat(2, 1, 1, 1)
    res, calls = {}, {}
    keep = (Q.FUSE_NORM_QUANT, Q.FUSE_NORM_QUANT_CLUSTERS, F.TENSOR_FOLD)
    counted = [F.RMSNormActQuant, F.GatherLayerNormActQuant]
    try:
        F.TENSOR_FOLD = False  # (narrow per-tensor slabs are otherwise reshaped by the separate path: other summation order)
        for fuse in (True, False):
            Q.FUSE_NORM_QUANT, Q.FUSE_NORM_QUANT_CLUSTERS = fuse, fuse
            n0 = sum(getattr(c, "_calls", 0) for c in counted)
            q = builder(d)
            q.set_quant_state(weight_quant=False, act_quant=True)
            q.let = True
            init_let(q, act_scales, shifts, prefix, cfg, None, torch.float32, torch.device(DEV), is_llama=is_llama)
            with torch.no_grad():
                q.float()
            losses = []
            for _step in range(2):
                q.zero_grad()
                q.smooth_and_quant_temporary()
                loss = torch.nn.functional.mse_loss(q(x, attention_mask=mask)[0], y)
                loss.backward()
                losses.append(loss.item())
                for aq in q.act_quantizers():
                    aq.free()
            res[fuse] = (losses, {n: p.grad.clone() for n, p in q.named_parameters() if p.grad is not None},
                         [aq.scale.clone() for aq in q.act_quantizers() if aq.scale is not None])
            calls[fuse] = sum(getattr(c, "_calls", 0) for c in counted) - n0
    finally:
        Q.FUSE_NORM_QUANT, Q.FUSE_NORM_QUANT_CLUSTERS, F.TENSOR_FOLD = keep
    assert res[True][0] == res[False][0]
    for a, b in zip(res[True][2], res[False][2]):
        assert torch.equal(a, b)
    assert res[True][1].keys() == res[False][1].keys()
    for n, g in res[True][1].items():
        assert rel_err(g, res[False][1][n]) <= 2e-5, n
    assert calls[True] == 4 and calls[False] == 0  # two norms x two steps went through the fused pair
