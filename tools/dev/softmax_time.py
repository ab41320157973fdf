"""Development timing of the attention-score pass at the Llama-2-7B score shape: forward and backward, with an fp32
mask, an fp16 mask and no mask (how much of the time is the mask's)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "llm-quantization_amd"))
from miq import functional as F  # noqa: E402

dev = "cuda:0"
B, H, Q, K = 1, 32, 2048, 2048
g = torch.Generator(device=dev).manual_seed(0)
scores = (torch.randn(B, H, Q, K, generator=g, device=dev) * 4).half()
go = torch.randn(B, H, Q, K, generator=g, device=dev).half()
m32 = torch.full((Q, K), torch.finfo(torch.float32).min, device=dev).triu(1).view(1, 1, Q, K)
m16 = torch.full((Q, K), torch.finfo(torch.float16).min, device=dev, dtype=torch.float16).triu(1).view(1, 1, Q, K)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


z32 = torch.zeros_like(m32)
for name, mask in (("fp32 mask", m32), ("fp16 mask", m16), ("zero mask", z32), ("no mask", None)):
    xs = scores.clone().requires_grad_(True)
    out = F.MaskedSoftmax.apply(xs, mask, math.sqrt(128.0))
    t_f = timeit(lambda: F.MaskedSoftmax.apply(scores, mask, math.sqrt(128.0)))
    t_b = timeit(lambda: torch.autograd.grad(out, xs, go, retain_graph=True))
    print(f"{name:10s} forward {t_f:7.1f} us   backward {t_b:7.1f} us")
