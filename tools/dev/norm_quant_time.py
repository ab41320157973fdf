#!/usr/bin/env python3
"""Times norm -> activation quantizer (fresh statistics) as two functions against the fused pair, forward and forward +
backward, HIP-graph replay over rotating buffers: python3 tools/dev/norm_quant_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "llm-quantization_amd"))
from miq import functional as F  # noqa: E402

DEV = "cuda:0"


def time_graph(fn, reps=9):
    # warm up ON the capture stream: per-stream state of the library (the activation kernels' statistics table) must
    # exist before the capture, which refuses to allocate it
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        fn()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    R = 2048
    for C, dt, counts, kind in ((4096, torch.float32, [128] * 32, "rms"), (4096, torch.float16, [128] * 32, "rms"),
                                (4096, torch.float32, None, "rms"), (4096, torch.float16, None, "rms"),
                                (4096, torch.float16, [128] * 32, "ln"), (4096, torch.float32, [128] * 32, "ln")):
        layout = F.ClusterLayout.get(counts, C, DEV)
        nbuf = 12
        xs = [torch.randn(R, C, device=DEV).to(dt) for _ in range(nbuf)]
        w = torch.ones(C, device=DEV)
        idx = torch.randperm(C, device=DEV).to(torch.int32)
        inv = torch.argsort(idx.long()).to(torch.int32)
        g = torch.randn(R, C, device=DEV).to(dt) * 1e-3
        out_dt = dt

        def sep(x, bwd):
            xa = x.detach().requires_grad_(bwd)
            if kind == "rms":
                y = F.RMSNorm.apply(xa, w, None, 1e-6)
            else:
                y = F.GatherLayerNorm.apply(xa, idx, inv, w.to(dt), None, 1e-5, out_dt)
            q = F.ActCalibFQ.apply(y, 4, "minmax", layout, None, None, {})
            if bwd:
                torch.autograd.grad(q, [xa], g)

        def fused(x, bwd):
            xa = x.detach().requires_grad_(bwd)
            if kind == "rms":
                q = F.RMSNormActQuant.apply(xa, w, None, 1e-6, 4, layout, {})
            else:
                q = F.GatherLayerNormActQuant.apply(xa, idx, w.to(dt), None, 1e-5, out_dt, 4, layout, {})
            if bwd:
                torch.autograd.grad(q, [xa], g)

        for bwd in (False, True):
            t_sep = time_graph(lambda: [sep(x, bwd) for x in xs]) / nbuf
            t_fus = time_graph(lambda: [fused(x, bwd) for x in xs]) / nbuf
            print(f"{kind} [{R},{C}] {str(dt)[6:]:8s} {'per-tensor' if counts is None else 'clusters of %d' % counts[0]:16s} "
                  f"{'fwd+bwd' if bwd else 'fwd    '}  separate {t_sep:7.1f} us   fused {t_fus:7.1f} us")


if __name__ == "__main__":
    main()
