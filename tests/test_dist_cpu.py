"""Data-parallel plumbing on CPU: two gloo ranks exercise the packed gradient all-reduce, the
[-min, max] statistics all-reduce, sample/row sharding and the rank-consistent optimizer step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from miq import dist as D
from miq import reorder_utils as R
from miq.calibrate import LossScaler, sync_reorder_stats


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def run2(fn, world=2):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return [ret[r] for r in range(world)]


def test_shard_range_partitions_everything():
    for n in (0, 1, 7, 128, 129):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                a, b = D.shard_range(n, r, world)
                seen.extend(range(a, b))
                for j in range(a, b):
                    assert D.owner_of(j, n, world) == r
            assert seen == list(range(n))


def _grads(rank, world):
    ctx = D.DPContext.from_env()
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.zeros(5, 1)), torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(3))]
    ps[0].grad = torch.full((5, 1), float(rank + 1))
    ps[1].grad = torch.arange(7.0) * (rank + 1)
    # ps[2] has a gradient on rank 1 only
    if rank == 1:
        ps[2].grad = torch.ones(3) * 4
    D.allreduce_mean_grads(ps, ctx)
    return [p.grad.clone() for p in ps]


def test_packed_gradient_allreduce():
    out = run2(_grads)
    for r in range(2):
        assert torch.equal(out[r][0], torch.full((5, 1), 1.5))
        assert torch.equal(out[r][1], torch.arange(7.0) * 1.5)
        assert torch.equal(out[r][2], torch.ones(3) * 2)


def _minmax(rank, world):
    ctx = D.DPContext.from_env(sync_stats=True)
    g = torch.Generator().manual_seed(rank)
    x = torch.randn(16, 12, generator=g)
    cmin, cmax = x.amin(0), x.amax(0)
    D.allreduce_minmax(cmin, cmax, ctx)
    return cmin, cmax


def test_minmax_allreduce_equals_global_batch():
    out = run2(_minmax)
    xs = [torch.randn(16, 12, generator=torch.Generator().manual_seed(r)) for r in range(2)]
    full = torch.cat(xs)
    for r in range(2):
        assert torch.equal(out[r][0], full.amin(0)) and torch.equal(out[r][1], full.amax(0))


def _reorder_stats(rank, world):
    ctx = D.DPContext.from_env()
    R.oc_maxmin_dict.clear()
    R.ic_maxmin_dict.clear()
    g = torch.Generator().manual_seed(10 + rank)
    R.oc_maxmin_dict["a"] = (torch.rand(8, generator=g), -torch.rand(8, generator=g))
    R.ic_maxmin_dict["b"] = (torch.rand(4, generator=g).half(), -torch.rand(4, generator=g).half())
    sync_reorder_stats(ctx)
    return R.oc_maxmin_dict["a"], R.ic_maxmin_dict["b"]


def test_reorder_stats_sync():
    out = run2(_reorder_stats)
    assert torch.equal(out[0][0][0], out[1][0][0]) and torch.equal(out[0][0][1], out[1][0][1])
    assert out[0][1][0].dtype == torch.float16
    gs = [torch.Generator().manual_seed(10 + r) for r in range(2)]
    mx = [torch.rand(8, generator=g) for g in gs]
    assert torch.equal(out[0][0][0], torch.max(mx[0], mx[1]))


def _train(rank, world):
    """Each rank sees one sample; after the averaged step both ranks hold the parameters of a
    single-process step on the two-sample batch."""
    ctx = D.DPContext.from_env()
    torch.manual_seed(3)
    lin = torch.nn.Linear(6, 4)
    x = torch.randn(2, 6)
    y = torch.randn(2, 4)
    opt = torch.optim.AdamW(lin.parameters(), lr=1e-2, weight_decay=0.0)
    scaler = LossScaler("cpu")
    for _ in range(3):
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(lin(x[rank:rank + 1]), y[rank:rank + 1])
        scaler(loss, opt, lin.parameters(), ctx)
    return [p.detach().clone() for p in lin.parameters()]


def test_dp_step_matches_big_batch():
    out = run2(_train)
    torch.manual_seed(3)
    lin = torch.nn.Linear(6, 4)
    x = torch.randn(2, 6)
    y = torch.randn(2, 4)
    opt = torch.optim.AdamW(lin.parameters(), lr=1e-2, weight_decay=0.0)
    for _ in range(3):
        opt.zero_grad()
        torch.nn.functional.mse_loss(lin(x), y).backward()
        opt.step()
    for r in range(2):
        for p, q in zip(out[r], lin.parameters()):
            assert torch.allclose(p, q, atol=1e-6)
    for p, q in zip(out[0], out[1]):
        assert torch.equal(p, q)


def _rows(rank, world):
    ctx = D.DPContext.from_env()
    w = [torch.arange(10 * 3.0).reshape(10, 3), torch.arange(7 * 2.0).reshape(7, 2)]
    return [(s.start, s.stop, v.clone()) for s, v in D.shard_weight_rows(w, ctx)]


def test_weight_row_shards_cover_all_rows():
    out = run2(_rows)
    w0 = torch.arange(10 * 3.0).reshape(10, 3)
    got = torch.cat([out[r][0][2] for r in range(2)])
    assert torch.equal(got, w0)
    assert out[0][1][0] == 0 and out[1][1][1] == 7
