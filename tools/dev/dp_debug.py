"""Debug aid: parameter gradients of ONE training step, two gloo ranks on cuda:0 (batch 1 each, synchronised
statistics) against one process at batch size 2."""
import os, sys, socket
import torch, torch.distributed as dist, torch.multiprocessing as mp
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(REPO, "llm-quantization_amd"), REPO, os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
DEV = "cuda:0"; PREFIX = "model.layers.0"
ABITS = int(os.environ.get("ABITS", "4"))

LOG = []
def _patch():
    from miq import functional as F
    orig = F.ActCalibFQ.backward
    def logged(ctx, g):
        r = orig(ctx, g)
        if os.environ.get("DUMP") and len(LOG) == int(os.environ.get("CALL", "5")) - 1:
            torch.save({"x": ctx.saved_tensors[0].reshape(g.shape).cpu(), "g": g.cpu(), "dx": r[0].cpu()}, os.path.join(REPO, "gpurun_out", os.environ["DUMP"]))
        LOG.append((tuple(g.shape), g.double().abs().sum().item(), r[0].double().abs().sum().item(), (r[0].double() - g.double()).abs().sum().item()))
        return r
    F.ActCalibFQ.backward = staticmethod(logged)
    orig_bwd = F.colquant_bwd
    CNT = [0]
    def bwd_logged(g2d, x2d, scale, rzp, n_bits, route=None, dp=None):
        dx = orig_bwd(g2d, x2d, scale, rzp, n_bits, route=route, dp=dp)
        CNT[0] += 1
        if dp is not None and CNT[0] == int(os.environ.get("CALL", "5")):
            torch.cuda.synchronize()
            torch.save({"x": x2d.cpu(), "g": g2d.cpu(), "dx": dx.cpu(), "scale": scale.cpu(), "rzp": rzp.cpu(), "chan_min": route[0].cpu(),
                        "chan_max": route[1].cpu(), "kmin": route[2].cpu(), "kmax": route[3].cpu()},
                       os.path.join(REPO, "gpurun_out", f"v_dp_rank{dp.rank}.pt"))
        return dx
    F.colquant_bwd = bwd_logged

def one_step(d, xs, dp, batch):
    _patch()
    from miq.blocks import QuantLlamaBlock, quant_args
    from miq.calibrate import CalibConfig, init_let, _set_dp
    from miq.dist import allreduce_mean_grads
    m = d["meta"]
    w = {n: d[f"w_{n}"].to(DEV) for n in ("q", "k", "v", "o", "gate", "up", "down", "ln1_w", "ln2_w")}
    args = quant_args(wbits=4, abits=ABITS, lwc=True, a_dynamic_method="per_tensor", metric="ema_minmax", let=True)
    q = QuantLlamaBlock(w, m["heads"], args, eps=m["eps"], rope_theta=m["theta"]).to(DEV)
    cfg = CalibConfig(epochs=1, nsamples=4, batch_size=batch, let=True, lwc=True, deactive_amp=True, autocast_passes=False)
    act_scales = {f"{PREFIX}.{n}": d[f"act_scale_{n}"] for n in ("self_attn.q_proj", "self_attn.o_proj", "mlp.up_proj")}
    mask = d["mask"].to(DEV)
    q.set_quant_state(False, False)
    with torch.no_grad():
        q.float()
        y = torch.stack([q(x.unsqueeze(0), attention_mask=mask)[0][0] for x in xs])
    q.set_quant_state(False, True); q.let = True
    init_let(q, act_scales, {}, PREFIX, cfg, None, torch.float32, torch.device(DEV), True)
    _set_dp(q, dp)
    q.smooth_and_quant_temporary()
    out = q(xs, attention_mask=mask.repeat(batch, 1, 1, 1))[0]
    loss = torch.nn.functional.mse_loss(y, out)
    loss.backward()
    params = list(q.omni_parameters(True))
    if dp is not None and dp.active:
        allreduce_mean_grads(params, dp)
    names = [n for n, _ in q.named_parameters() if ("smooth" in n or "bound_factor" in n)]
    return float(loss), {n: p.grad.detach().float().cpu().clone() for n, p in zip(names, params)}, list(LOG)

def rank_main(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from miq.dist import DPContext
    from helpers import golden
    d = golden("block_llama")["llama_tiny_w4a4_pertensor"]
    ctx = DPContext.from_env(sync_stats=True)
    ret[rank] = one_step(d, d["inps"][rank:rank + 1].to(DEV), ctx, 1)
    dist.destroy_process_group()

if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn"); ret = ctx.Manager().dict()
    ps = [ctx.Process(target=rank_main, args=(r, 2, port, ret)) for r in range(2)]
    [p.start() for p in ps]; [p.join(200) for p in ps]
    from helpers import golden
    d = golden("block_llama")["llama_tiny_w4a4_pertensor"]
    l1, g1, log1 = one_step(d, d["inps"][:2].to(DEV), None, 2)
    (la, ga, loga), (lb, gb, logb) = ret[0], ret[1]
    for e1, ea, eb in zip(log1, loga, logb):
        print(e1[0], "g %.6e vs %.6e | dx %.6e vs %.6e | route %.6e vs %.6e" % (2 * e1[1], ea[1] + eb[1], 2 * e1[2], ea[2] + eb[2], 2 * e1[3], ea[3] + eb[3]))
    print("loss single", l1, "dp mean", (la + lb) / 2)
    for n in g1:
        ref = g1[n]; got = ga[n]
        print(f"{n:60s} rel {((got-ref).norm()/ (ref.norm()+1e-30)).item():.3e}  ranks_equal {torch.equal(ga[n], gb[n])}")
