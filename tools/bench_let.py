#!/usr/bin/env python3
"""Micro-benchmark of the LET-fused weight quantizer kernels on Llama-2-7B shapes."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llm-quantization_amd"))
from miq import functional as F

def timeit(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

dev = "cuda:0"
for (O, I) in ((4096, 4096), (11008, 4096), (4096, 11008), (5120, 5120), (8192, 8192)):
    w = torch.randn(O, I, device=dev) * 0.02
    g = torch.randn(O, I, device=dev) * 1e-3
    cs = torch.rand(I, device=dev) + 0.5; rs = torch.rand(O, device=dev) + 0.5; sh = torch.randn(I, device=dev) * 0.1
    up = torch.full((O, 1), 0.982, device=dev); low = torch.full((O, 1), 0.982, device=dev); gb = torch.randn(O, device=dev)
    tf = timeit(lambda: F.let_rowquant_fwd(w, cs, rs, True, sh, up, low, 4, False))
    tb = timeit(lambda: F.let_rowquant_bwd(g, gb, w, cs, rs, True, sh, up, low, 4, False))
    tu = timeit(lambda: F.rowquant_uniform_bwd(g, w, 4, False, up, low))
    n = O * I * 4
    print(f"[{O}x{I}] let_fwd {tf:7.1f} us ({2*n/tf/1e3:6.0f} GB/s)  let_bwd {tb:7.1f} us ({2*n/tb/1e3:6.0f} GB/s)  plain uniform_bwd {tu:7.1f} us ({3*n/tu/1e3:6.0f} GB/s)")
    # the forms of a step under fp16 autocast: fp16 quantized weight out, fp16 weight gradient in; row factor multiplied
    # (q projection) and column factor only (up / gate projections)
    gh = g.half()
    for name, r, rdiv in (("cs*rs", rs, False), ("cs/rs", rs, True), ("cs only", None, False)):
        tf = timeit(lambda: F.let_rowquant_fwd(w, cs, r, rdiv, sh, up, low, 4, False, out_dtype=torch.float16))
        tb = timeit(lambda: F.let_rowquant_bwd(gh, gb, w, cs, r, rdiv, sh, up, low, 4, False))
        print(f"    fp16 out / grad, {name:8s} let_fwd {tf:7.1f} us ({1.5*n/tf/1e3:6.0f} GB/s)  let_bwd {tb:7.1f} us ({1.5*n/tb/1e3:6.0f} GB/s)")
