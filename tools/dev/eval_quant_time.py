#!/usr/bin/env python3
"""Quantize-dequantize with calibrated per-channel parameters (miq_colquant_fwd, the eval path of the static activation
quantizer) on the Llama-2-7B activation shapes: GB/s of 2ns over rotating buffers larger than the Infinity Cache."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_families import time_graph, F, DEV  # noqa: E402

T = 2048
for dt, es in ((torch.float32, 4), (torch.float16, 2)):
    for C in (4096, 11008):
        copies = max(3, int(600e6 // (T * C * es * 2)) + 1)
        xs = [torch.randn(T, C, device=DEV, dtype=dt) for _ in range(copies)]
        scale = (torch.rand(C, device=DEV) * 0.1 + 0.2).to(dt)
        rzp = torch.full((C,), 8.0, device=DEV, dtype=dt)
        for codes in (False, True):
            t = time_graph(lambda: [F.colquant_fwd(x, scale, rzp, 4, want_codes=codes) for x in xs], reps=15)
            n = T * C * es * copies
            print(f"{'f32' if es == 4 else 'f16'} [{T}, {C}] codes={int(codes)}: {t / copies * 1e6:.1f} us per tensor, {2 * n / t / 1e9:.0f} GB/s "
                  f"({2 * n / t / 8e12:.3f} of peak)", flush=True)
        del xs
        torch.cuda.empty_cache()
