"""Two gloo ranks on one GPU: all_reduce of CUDA tensors of several sizes, directly and through a host copy.
Run: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/dev/gloo_cuda_check.py"""
import datetime
import os
import sys
import time

import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=20))
torch.cuda.set_device(0)
for n in (1024, 84992, 1 << 20):
    for mode in ("host", "cuda"):
        t = torch.full((n,), float(rank + 1), device="cuda")
        torch.cuda.synchronize()
        t0 = time.time()
        try:
            if mode == "host":
                h = t.cpu()
                dist.all_reduce(h)
                t.copy_(h)
            else:
                dist.all_reduce(t)
            torch.cuda.synchronize()
            ok = bool((t == 3).all())
            print(f"rank {rank} n={n} {mode}: ok={ok} {1e3 * (time.time() - t0):.1f} ms", flush=True)
        except Exception as e:
            print(f"rank {rank} n={n} {mode}: FAILED {type(e).__name__}: {str(e)[:120]}", flush=True)
            sys.exit(1)
dist.destroy_process_group()
