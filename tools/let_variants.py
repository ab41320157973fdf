#!/usr/bin/env python3
"""Where the LET-fused forward kernel spends its time: the same [4096, 4096] call with parts of the work removed
(no shift dot product, product instead of quotient by the row factor, no column factor, plain quantizer)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llm-quantization_amd"))
from miq import functional as F

def timeit(fn, n=30):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

dev = "cuda:0"
O, I = 4096, 4096
w = torch.randn(O, I, device=dev) * 0.02
cs = torch.rand(I, device=dev) + 0.5; rs = torch.rand(O, device=dev) + 0.5; sh = torch.randn(I, device=dev) * 0.1
up = torch.full((O, 1), 0.982, device=dev); low = torch.full((O, 1), 0.982, device=dev)
for half in (False, True):
    od = torch.float16 if half else None
    print("half out" if half else "fp32 out")
    print("  full            %.1f us" % timeit(lambda: F.let_rowquant_fwd(w, cs, rs, True, sh, up, low, 4, False, out_dtype=od)))
    print("  no shift        %.1f us" % timeit(lambda: F.let_rowquant_fwd(w, cs, rs, True, None, up, low, 4, False, out_dtype=od)))
    print("  rs as product   %.1f us" % timeit(lambda: F.let_rowquant_fwd(w, cs, rs, False, sh, up, low, 4, False, out_dtype=od)))
    print("  no cs           %.1f us" % timeit(lambda: F.let_rowquant_fwd(w, None, rs, True, sh, up, low, 4, False, out_dtype=od)))
    print("  no factors      %.1f us" % timeit(lambda: F.let_rowquant_fwd(w, None, None, True, None, up, low, 4, False, out_dtype=od)))
print("plain uniform fwd %.1f us" % timeit(lambda: F.rowquant_uniform_fwd(w, 4, False, up, low)))
