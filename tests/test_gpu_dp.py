"""Data-parallel block calibration executed for real: two processes share cuda:0 and talk over gloo (RCCL refuses two
ranks on one device; the collectives and their placement are the same code path).  Each rank holds every second sample of
the tiny Llama fixture and runs calibrate_layer with batch 1 and synchronised activation statistics; the reference
semantics it must reproduce is ONE process at --batch_size 2 on the same samples in the same order
(quantize/omniquant.py:502-524: batch j = samples 2j, 2j+1; mean loss over the batch; statistics over the whole batch).

Checked: both ranks end with bit-identical parameters and activation scales; their per-step losses average to the
single-process loss; the learned parameters and the propagated activations agree with the single-process run to the
tolerance GEMMs of another batch shape allow (they flip single quantization codes, see test_gpu_block_llama.py).
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PREFIX = "model.layers.0"
LET_SITES = ("self_attn.q_proj", "self_attn.o_proj", "mlp.up_proj")


def _setup():
    from test_gpu_block_llama import build, config
    d = golden("block_llama")["llama_tiny_w4a4_pertensor"]
    act_scales = {f"{PREFIX}.{n}": d[f"act_scale_{n}"] for n in LET_SITES}
    return d, build, config, act_scales


def _scales(q):
    out = {}
    for name, mod in q.named_modules():
        aq = getattr(mod, "act_quantizer", None)
        if aq is not None and getattr(aq, "scale", None) is not None:
            out[name] = aq.scale.detach().float().cpu().reshape(-1).clone()
    return out


def _rank_main(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from miq.calibrate import calibrate_layer
        from miq.dist import DPContext, sample_indices
        torch.cuda.set_device(0)
        d, build, config, act_scales = _setup()
        m = d["meta"]
        ctx = DPContext.from_env(sync_stats=True)
        mine = list(sample_indices(m["N"], rank, world))
        q = build(d)
        quant = d["inps"][mine].clone().to(DEV)
        fp = d["inps"][mine].clone().to(DEV)
        cfg = config(m)  # batch_size 1 per rank
        res = calibrate_layer(q, quant, fp, d["mask"].to(DEV), cfg, act_scales, {}, PREFIX, is_llama=True, dp=ctx)
        ret[rank] = {"losses": list(res["losses"]), "state": {k: v.float().cpu() for k, v in res["omni_state"].items()},
                     "quant": quant.float().cpu(), "mine": mine, "scales": _scales(q)}
    finally:
        dist.destroy_process_group()


def test_two_ranks_equal_one_process_at_batch_size_two():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    a, b = ret[0], ret[1]
    # the ranks never diverge
    for k in a["state"]:
        assert torch.equal(a["state"][k], b["state"][k]), k
    assert a["scales"].keys() == b["scales"].keys() and len(a["scales"]) >= 4
    for k in a["scales"]:
        assert torch.equal(a["scales"][k], b["scales"][k]), k

    # one process, batch size 2, same samples in the same order
    from miq.calibrate import calibrate_layer
    d, build, config, act_scales = _setup()
    m = d["meta"]
    q = build(d)
    cfg = config(m)
    cfg.batch_size = 2
    quant, fp = d["inps"].clone().to(DEV), d["inps"].clone().to(DEV)
    res = calibrate_layer(q, quant, fp, d["mask"].to(DEV), cfg, act_scales, {}, PREFIX, is_llama=True)
    one = torch.tensor(res["losses"])
    two = (torch.tensor(a["losses"]) + torch.tensor(b["losses"])) / 2
    assert one.shape == two.shape == (m["epochs"] * m["N"] // 2,)
    assert torch.allclose(one, two, rtol=2e-2), (one, two)
    assert abs(float(one[0]) - float(two[0])) <= 1e-5 * float(one[0])  # first step: same parameters, same statistics
    for k, ref in res["omni_state"].items():
        got, ref = a["state"][k], ref.float().cpu()
        err = (got - ref).abs()
        assert (err <= 2e-3 + 1e-3 * ref.abs().max().item()).float().mean().item() >= 0.98, k
        assert err.max().item() <= 4e-2, k
    full = torch.empty_like(quant.float().cpu())
    full[a["mine"]] = a["quant"]
    full[b["mine"]] = b["quant"]
    assert rel_err(full, quant.float().cpu()) <= 5e-2
    assert ((full - quant.float().cpu()).abs().mean() / quant.float().cpu().abs().mean()).item() <= 2e-2
