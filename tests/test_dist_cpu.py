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
from miq.calibrate import CalibConfig, LossScaler, calibrate_layer, sync_reorder_stats


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


def test_sample_shards_are_dealt_round_robin():
    """Rank r holds samples r, r + W, ...: local step j of the W ranks is the reference's batch j at --batch_size W."""
    for n in (0, 1, 7, 128, 129):
        for world in (1, 2, 3, 8):
            seen = {}
            for r in range(world):
                for at, j in enumerate(D.sample_indices(n, r, world)):
                    assert D.sample_owner(j, world) == (r, at)
                    seen[j] = r
            assert sorted(seen) == list(range(n))
    assert [list(D.sample_indices(8, r, 2))[1] for r in range(2)] == [2, 3]  # step 1 of two ranks = samples {2, 3}


class _StubBlock(torch.nn.Module):
    """The part of a quantized block's interface that calibrate_layer drives, on plain CPU ops: one linear layer whose
    weight is scaled by a trainable per-row factor (standing in for the clipping factors)."""

    def __init__(self, h):
        super().__init__()
        g = torch.Generator().manual_seed(5)
        self.weight = torch.nn.Parameter(torch.randn(h, h, generator=g) * 0.3, requires_grad=False)
        self.upbound_factor = torch.nn.Parameter(torch.ones(h, 1))
        self.calls = []

    def forward(self, x, attention_mask=None, position_ids=None):
        self.calls.append(float(x.sum()))
        w = self.weight * self.upbound_factor if self.training_mode else self.weight
        return (torch.tanh(x @ w.t()),)

    training_mode = False
    let = False

    def set_quant_state(self, weight_quant=False, act_quant=False):
        self.training_mode = act_quant

    def smooth_and_quant_temporary(self): pass
    def smooth_and_quant_inplace(self): pass
    def clear_temp_variable(self): pass
    def register_scales_and_zeros(self): pass
    def act_quantizers(self): return []
    def let_parameters(self, use_shift=True): return []
    def lwc_parameters(self): return [self.upbound_factor]
    def omni_parameters(self, use_shift=True): return [self.upbound_factor]
    def omni_state_dict(self): return {"upbound_factor": self.upbound_factor.detach().clone()}


def _samples(n, t, h):
    return torch.randn(n, t, h, generator=torch.Generator().manual_seed(11))


def _uneven(rank, world):
    """7 samples over 3 ranks (3 + 2 + 2): every rank must take the same number of optimizer steps and of broadcasts."""
    ctx = D.DPContext.from_env()
    n, t, h = 7, 4, 6
    full = _samples(n, t, h)
    mine = list(D.sample_indices(n, rank, world))
    block = _StubBlock(h)
    quant, fp = full[mine].clone(), full[mine].clone()
    cfg = CalibConfig(epochs=2, nsamples=n, batch_size=1, let=False, lwc=True, deactive_amp=True)
    res = calibrate_layer(block, quant, fp, None, cfg, dp=ctx)
    return {"steps": len(res["losses"]), "param": block.upbound_factor.detach().clone(), "quant": quant, "mine": mine}


def test_uneven_sample_shards_do_not_deadlock_and_stay_consistent():
    out = run2(_uneven, world=3)
    assert [o["steps"] for o in out] == [4, 4, 4]  # 2 epochs x (7 // 3) steps on every rank
    for o in out[1:]:
        assert torch.equal(o["param"], out[0]["param"])
    # the propagation pass visited the samples in global order and every rank kept the outputs of its own samples:
    # the same outputs as one process running the trained block over all seven samples
    block = _StubBlock(6)
    with torch.no_grad():
        block.upbound_factor.copy_(out[0]["param"])
    block.set_quant_state(False, True)
    full = _samples(7, 4, 6)
    with torch.no_grad():
        want = torch.stack([block(full[j:j + 1])[0][0] for j in range(7)])
    for o in out:
        assert torch.allclose(o["quant"], want[o["mine"]], atol=1e-6)


def _too_few(rank, world):
    ctx = D.DPContext.from_env()
    full = _samples(4, 4, 6)
    block = _StubBlock(6)
    mine = full[:1].clone()  # one sample although the share of 4 samples over 2 ranks is two
    cfg = CalibConfig(epochs=1, nsamples=4, batch_size=1, let=False, lwc=True, deactive_amp=True)
    try:
        calibrate_layer(block, mine, mine.clone(), None, cfg, dp=ctx)
    except ValueError as e:
        return str(e)
    return "no error"


def test_a_rank_with_too_few_samples_is_refused_before_any_collective():
    out = run2(_too_few)
    assert all("fewer than its share" in o for o in out)



def _stub_quantize(w):
    """A row-wise stand-in for the weight quantizers (4-bit min/max grid per row), plain torch so that it runs on CPU."""
    mn, mx = w.amin(1, keepdim=True), w.amax(1, keepdim=True)
    s = ((mx - mn) / 15).clamp(1e-5, 1e4)
    z = (-mn / s).round()
    return ((w / s).round() + z).clamp(0, 15).sub(z).mul(s)


def _sharded_quant(rank, world):
    ctx = D.DPContext.from_env()
    g = torch.Generator().manual_seed(11)
    weights = [torch.randn(o, i, generator=g) for o, i in ((8, 5), (7, 3), (12, 4), (2, 6))]  # even and uneven row counts
    full = D.quantize_rows_sharded(weights, ctx, _stub_quantize, gather=True)
    parts = D.quantize_rows_sharded(weights, ctx, _stub_quantize, gather=False)
    rows = [(sl.start, sl.stop) for sl, _q in parts]
    return [f.clone() for f in full], rows


@pytest.mark.parametrize("world", [2, 3])
def test_row_sharded_quantization_reassembles_the_full_matrices(world):
    """shard by output channel -> quantize the shard -> all-gather == quantizing the whole matrix (rows are independent);
    row counts that do not divide by the world size go through the padded gather."""
    res = run2(_sharded_quant, world)
    g = torch.Generator().manual_seed(11)
    weights = [torch.randn(o, i, generator=g) for o, i in ((8, 5), (7, 3), (12, 4), (2, 6))]
    want = [_stub_quantize(w) for w in weights]
    for r in range(world):
        full, rows = res[r]
        for f, w in zip(full, want):
            assert torch.equal(f, w)
        for (a, b), w in zip(rows, weights):
            assert (a, b) == D.shard_range(w.shape[0], r, world)
