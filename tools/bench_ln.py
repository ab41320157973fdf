import sys, torch
sys.path.insert(0, "llm-quantization_amd")
from miq import functional as F
def timeit(fn, n=50):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
dev = "cuda:0"
for dt in (torch.float16, torch.float32):
    for (R, C) in ((2048, 4096), (2048, 11008)):
        x = torch.randn(R, C, device=dev, dtype=dt)
        w = torch.rand(C, device=dev, dtype=dt) + 0.5; b = torch.randn(C, device=dev, dtype=dt)
        idx = torch.randperm(C, device=dev).to(torch.int32)
        tf = timeit(lambda: F.gather_layernorm_fwd(x, idx, w, b, 1e-5))
        tu = timeit(lambda: torch.nn.functional.layer_norm(F.gather_lastdim(x, idx), (C,), w, b, 1e-5))
        tl = timeit(lambda: torch.nn.functional.layer_norm(x, (C,), w, b, 1e-5))
        n = R * C * x.element_size()
        print(f"{dt} [{R}x{C}] fused {tf:6.1f} us ({2*n/tf/1e3:6.0f} GB/s) | gather + torch LN {tu:6.1f} us | torch LN alone {tl:6.1f} us")
