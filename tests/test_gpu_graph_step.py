"""A calibration step replayed from one HIP graph (CalibConfig.graph_step) must follow the eager trajectory: same
gradients, hence bit-identical parameters and propagated activations; the loss values agree to summation order (the
graph path takes the loss kernel instead of ATen's reduction for fp32 operands).  ATen's multi-block reductions do not
survive graph replay on this ROCm build, so no captured step may contain one: the reordered OPT block used to route the
column sums of its smoothing-scale gradients through one (its replays were caught by the loop's built-in comparison
and it fell back to eager steps); they go through this library's column-sum kernel now, and the block must REPLAY."""
import pytest
import torch

from helpers import golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PREFIX = "model.layers.0"
LET_SITES = ("self_attn.q_proj", "self_attn.o_proj", "mlp.up_proj")


@pytest.mark.parametrize("amp", [False, True])
def test_graph_replayed_steps_equal_eager_steps(amp):
    from test_gpu_block_llama import build, config
    from miq.calibrate import calibrate_layer
    d = golden("block_llama")["llama_tiny_w4a4_pertensor"]
    m = d["meta"]
    act_scales = {f"{PREFIX}.{n}": d[f"act_scale_{n}"] for n in LET_SITES}
    out = {}
    for graph in (False, True):
        q = build(d)
        cfg = config(m)
        cfg.deactive_amp = not amp
        cfg.graph_step = graph
        cfg.epochs = 6
        quant, fp = d["inps"].clone().to(DEV), d["inps"].clone().to(DEV)
        res = calibrate_layer(q, quant, fp, d["mask"].to(DEV), cfg, act_scales, {}, PREFIX, is_llama=True)
        out[graph] = (res["losses"], {k: v.clone() for k, v in res["omni_state"].items()}, quant.clone(), res["graph_steps"])
    steps = 6 * m["N"]
    assert len(out[True][0]) == steps
    assert out[False][3] == 0 and out[True][3] == steps - 3  # one eager step, two compared steps, the rest replayed (< 64 steps)
    # (the graph path always takes the loss kernel, the eager path ATen's reduction for fp32 operands: same gradients,
    # the reported value in another summation order)
    assert torch.allclose(torch.tensor(out[True][0]), torch.tensor(out[False][0]), rtol=1e-6, atol=0)
    for k, v in out[False][1].items():
        assert torch.equal(v, out[True][1][k]), k
    assert torch.equal(out[True][2], out[False][2])


def test_the_reordered_opt_block_replays_too():
    import test_gpu_block_calibration as opt
    from miq import reorder_utils as R
    from miq.calibrate import CalibConfig, calibrate_layer, reorder_block
    d = golden("block")["opt_tiny_w4a4_reorder"]
    m = d["meta"]
    prefix = "model.decoder.layers.0"
    sites = ("self_attn.q_proj", "self_attn.out_proj", "fc1")
    act_scales = {f"{prefix}.{n}": d[f"act_scale_{n}"] for n in sites}
    act_shifts = {f"{prefix}.{n}": d[f"act_shift_{n}"] for n in sites}
    out = {}
    for graph in (False, True):
        q = opt.build(d)
        R.oc_maxmin_dict.clear()
        R.ic_maxmin_dict.clear()
        for name in ("self_attn_layer_norm", "self_attn.q_proj", "self_attn.k_proj", "final_layer_norm"):
            R.oc_maxmin_dict[name] = (d[f"stat_oc_{name}_max"].to(DEV), d[f"stat_oc_{name}_min"].to(DEV))
        for name in ("self_attn.out_proj", "fc2"):
            R.ic_maxmin_dict[name] = (d[f"stat_ic_{name}_max"].to(DEV), d[f"stat_ic_{name}_min"].to(DEV))
        indices = reorder_block(q, m["H"], m["F"], m["heads"], {f"R{k}": True for k in range(1, 6)}, m["clusters"], DEV)
        cfg = CalibConfig(epochs=m["epochs"], nsamples=m["N"], batch_size=1, let=True, lwc=True, deactive_amp=True,
                          autocast_passes=False, reorder="012345", n_clusters=m["clusters"], graph_step=graph)
        quant, fp = d["inps"].clone().to(DEV), d["inps"].clone().to(DEV)
        res = calibrate_layer(q, quant, fp, d["mask"].to(DEV), cfg, act_scales, act_shifts, prefix,
                              dims=(m["H"], m["F"], m["heads"]), indices=indices)
        R.oc_maxmin_dict.clear()
        R.ic_maxmin_dict.clear()
        out[graph] = (res["losses"], {k: v.clone() for k, v in res["omni_state"].items()}, quant.clone(), res["graph_steps"],
                      res["notes"])
    steps = m["epochs"] * m["N"]
    # one eager step, two steps run both ways and compared, every other step replayed; nothing given up on the way
    assert out[False][3] == 0 and out[True][3] == steps - 3 and out[True][4] == [], (out[True][3], out[True][4])
    assert torch.allclose(torch.tensor(out[True][0]), torch.tensor(out[False][0]), rtol=1e-6, atol=0)
    for k, v in out[False][1].items():
        assert torch.equal(v, out[True][1][k]), k
    assert torch.equal(out[True][2], out[False][2])
