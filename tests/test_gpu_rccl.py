"""The collectives of the data-parallel path on RCCL itself (backend "nccl"), with the one rank a one-GPU box allows:
the packed gradient all-reduce, the [-min, max] statistics all-reduce, the broadcast of the propagation pass and the
all-gather of row shards run through the library's own functions on GPU tensors and must return their inputs (a mean,
a maximum and a concatenation over one rank).  Multi-rank semantics are covered by the gloo tests on the CPU; this one
checks that every collective, reduction operator, dtype and tensor layout the path uses is one RCCL accepts."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, os.path.join(sys.argv[1], "llm-quantization_amd"))
    import torch
    import torch.distributed as dist
    from miq import dist as D

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))

    class Forced(D.DPContext):  # one rank, but take the collective path
        @property
        def active(self):
            return True

    ctx = Forced(dist.group.WORLD, 0, 1, True)
    g = torch.Generator(device="cuda").manual_seed(3)
    params = [torch.nn.Parameter(torch.randn(s, device="cuda", generator=g).to(dt))
              for s, dt in (((4096, 1), torch.float32), ((11008, 1), torch.float32), ((4096,), torch.float16), ((1,), torch.float32))]
    for p in params[:-1]:
        p.grad = torch.randn(p.shape, device="cuda", generator=g).to(p.dtype)
    want = [None if p.grad is None else p.grad.clone() for p in params]
    buf = D.allreduce_mean_grads(params, ctx)
    buf = D.allreduce_mean_grads(params, ctx, buf)  # the reused bucket
    for p, w in zip(params, want):
        w = torch.zeros_like(p) if w is None else w
        assert torch.equal(p.grad, w), (tuple(p.shape), p.dtype, float((p.grad.float() - w.float()).abs().max()))
    cmin = -torch.rand(4096, device="cuda", generator=g)
    cmax = torch.rand(4096, device="cuda", generator=g)
    a, b = cmin.clone(), cmax.clone()
    D.allreduce_minmax(a, b, ctx)
    assert torch.equal(a, cmin) and torch.equal(b, cmax)
    x = torch.randn(2048, 4096, device="cuda", generator=g).half()
    y = x.clone()
    D.broadcast_from(y, 0, ctx)
    assert torch.equal(x, y)
    for rows in (4096, 4095):
        w = torch.randn(rows, 512, device="cuda", generator=g)
        full = D.gather_row_shards(w, rows, ctx)
        assert torch.equal(full, w)
    full = D.quantize_rows_sharded([w], ctx, lambda t: t * 2)[0]
    assert torch.equal(full, w * 2)
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("RCCL_OK", torch.cuda.nccl.version())
""")


def test_collectives_of_the_dp_path_run_on_rccl(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "rccl_one_rank.py"
    script.write_text(SCRIPT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(script), REPO], env=env, capture_output=True, text=True, timeout=100)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
