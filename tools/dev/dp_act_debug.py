"""Debug aid: one static activation quantizer, rows split over two gloo ranks with synchronised statistics, against
one process on all rows."""
import os, sys, socket
import torch, torch.distributed as dist, torch.multiprocessing as mp
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(REPO, "llm-quantization_amd"), REPO, os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
DEV = "cuda:0"
SHAPES = [((2, 4, 32, 16), None), ((2, 32, 64), None), ((2, 32, 64), [16] * 4), ((2, 64, 256), None)]

def data(shape, seed=3):
    f = os.path.join(REPO, "gpurun_out", "v_single.pt")
    if shape == (2, 4, 32, 16) and os.path.exists(f):
        d = torch.load(f)
        return d["x"], d["g"]
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g), torch.randn(shape, generator=g) * 1e-2

def run(x, go, counts, dp):
    from miq import functional as F
    C = x.shape[-1]
    layout = F.ClusterLayout.get(counts, C, DEV)
    leaf = x.to(DEV).requires_grad_(True)
    h = {}
    out = F.ActCalibFQ.apply(leaf, 4, "ema_minmax", layout, None, None, h, dp)
    (dx,) = torch.autograd.grad(out, leaf, go.to(DEV))
    return out.detach().cpu(), dx.cpu(), h["scale"].float().cpu()

def rank_main(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from miq.dist import DPContext
    ctx = DPContext.from_env(sync_stats=True)
    res = []
    for shape, counts in SHAPES:
        x, go = data(shape)
        res.append(run(x[rank:rank + 1], go[rank:rank + 1], counts, ctx))
    ret[rank] = res
    dist.destroy_process_group()

if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn"); ret = ctx.Manager().dict()
    ps = [ctx.Process(target=rank_main, args=(r, 2, port, ret)) for r in range(2)]
    [p.start() for p in ps]; [p.join(200) for p in ps]
    for i, (shape, counts) in enumerate(SHAPES):
        x, go = data(shape)
        out, dx, sc = run(x, go, counts, None)
        o2 = torch.cat([ret[0][i][0], ret[1][i][0]]); d2 = torch.cat([ret[0][i][1], ret[1][i][1]])
        if i == 0 and os.path.exists(os.path.join(REPO, "gpurun_out", "v_single.pt")):
            ref = torch.load(os.path.join(REPO, "gpurun_out", "v_single.pt"))["dx"]
            print("single-process dx vs block dump", torch.equal(ref, dx), "dp vs dump", (ref - d2).abs().max().item(),
                  "x max at", [int(v) for v in torch.nonzero(x == x.max())[0]], "x min at", [int(v) for v in torch.nonzero(x == x.min())[0]],
                  "ties max", int((x == x.max()).sum()), "ties min", int((x == x.min()).sum()))
            bad = torch.nonzero(dx != d2)
            print("differing positions", bad[:6].tolist(), dx[dx != d2][:6].tolist(), d2[dx != d2][:6].tolist())
        print(shape, counts, "out equal", torch.equal(out, o2), "scale equal", torch.equal(sc, ret[0][i][2]),
              "dx rel", ((dx - d2).norm() / dx.norm()).item(), "n diff", int((dx != d2).sum()), "max|d|", (dx - d2).abs().max().item())
