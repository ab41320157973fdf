#!/usr/bin/env python3
"""Row-kernel micro-benchmark on working sets larger than the 256 MiB Infinity Cache (rotating buffers)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "llm-quantization_amd"))
from miq import functional as F

def timeit(fns, iters=40, warm=8):
    for i in range(warm):
        fns[i % len(fns)]()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fns[i % len(fns)]()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / iters

def main():
    dev = "cuda:0"
    for dt in (torch.float32, torch.float16):
        for (o, i) in [(4096, 4096), (11008, 4096), (4096, 11008), (8192, 8192)]:
            nbuf = max(2, int(600e6 // (o * i * (4 if dt == torch.float32 else 2))) + 1)
            xs = [torch.randn(o, i, device=dev, dtype=dt) * 0.02 for _ in range(nbuf)]
            gs = [torch.randn(o, i, device=dev, dtype=dt) for _ in range(nbuf)]
            up = torch.full((o, 1), 0.982, device=dev, dtype=dt); low = up.clone()
            nbytes = o * i * xs[0].element_size()
            g = torch.cuda.CUDAGraph()
            def run_f():
                for x in xs: F.rowquant_uniform_fwd(x, 4, False, up, low)
            def run_b():
                for x, gg in zip(xs, gs): F.rowquant_uniform_bwd(gg, x, 4, False, up, low, want_dx=True)
            for name, fn, mult in (("fwd", run_f, 2), ("bwd", run_b, 3)):
                fn(); torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g): fn()
                t = timeit([g.replay], iters=10, warm=3) / nbuf
                print(f"cfg={os.environ.get('MIQ_ROW_CFG','0')} {name} {str(dt)[6:]:8s} {o}x{i}: {t*1e6:8.1f} us  {mult*nbytes/t/1e9:8.1f} GB/s")
            del xs, gs
            torch.cuda.empty_cache()

if __name__ == "__main__":
    main()
