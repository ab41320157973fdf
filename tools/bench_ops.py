#!/usr/bin/env python3
"""Micro-benchmark of single kernels (GB/s of algorithmic bytes).  Development tool, not the judged bench."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "llm-quantization_amd"))
from miq import functional as F  # noqa: E402
from miq import synth  # noqa: E402


def timeit(fn, iters=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dev = "cuda:0"
    shapes = [(4096, 4096), (11008, 4096), (4096, 11008), (8192, 8192), (28672, 8192), (8192, 28672), (768, 768), (3072, 768)]
    for dt in (torch.float32, torch.float16):
        for (o, i) in shapes:
            x = torch.randn(o, i, device=dev, dtype=dt) * 0.02
            up = torch.full((o, 1), 0.982, device=dev, dtype=dt)
            low = torch.full((o, 1), 0.982, device=dev, dtype=dt)
            g = torch.randn_like(x)
            nbytes = x.numel() * x.element_size()
            t = timeit(lambda: F.rowquant_uniform_fwd(x, 4, False, up, low))
            print(f"uniform_fwd {str(dt)[6:]:8s} {o}x{i}: {t*1e6:8.1f} us  {2*nbytes/t/1e9:8.1f} GB/s")
            t = timeit(lambda: F.rowquant_uniform_bwd(g, x, 4, False, up, low, want_dx=True))
            print(f"uniform_bwd {str(dt)[6:]:8s} {o}x{i}: {t*1e6:8.1f} us  {3*nbytes/t/1e9:8.1f} GB/s")
            t = timeit(lambda: F.rowquant_uniform_bwd(g, x, 4, False, up, low, want_dx=False))
            print(f"uniform_bwd(no dx) {str(dt)[6:]:8s} {o}x{i}: {t*1e6:8.1f} us  {2*nbytes/t/1e9:8.1f} GB/s")
            y = torch.empty_like(x)
            t = timeit(lambda: y.copy_(x))
            print(f"torch copy  {str(dt)[6:]:8s} {o}x{i}: {t*1e6:8.1f} us  {2*nbytes/t/1e9:8.1f} GB/s")


if __name__ == "__main__":
    main()
