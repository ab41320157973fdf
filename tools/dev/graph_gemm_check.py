#!/usr/bin/env python3
"""Does a PyTorch-only fp32 (and fp16) linear forward + backward replayed from a HIP graph reproduce its eager results on
this ROCm build when small eager kernels run between the replays?  No kernel of this library is involved."""
import torch

DEV = "cuda:0"


def run(dtype, T, H, O, steps=12, interleave=True):
    g = torch.Generator(device=DEV).manual_seed(1)
    w = (torch.randn(O, H, device=DEV, generator=g) * 0.02).to(dtype).requires_grad_(True)
    b = torch.zeros(O, device=DEV, dtype=dtype, requires_grad=True)
    xs = [torch.randn(T, H, device=DEV, generator=g).to(dtype) for _ in range(steps)]

    def fb(x):
        y = torch.nn.functional.linear(x, w, b)
        loss = (y.float() ** 2).mean()
        loss.backward()
        return loss.detach()

    ref = []
    for x in xs:
        w.grad = b.grad = None
        l = fb(x)
        ref.append((l.clone(), w.grad.clone(), b.grad.clone()))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        w.grad = b.grad = None
        fb(xs[0])
    torch.cuda.current_stream().wait_stream(side)
    w.grad = b.grad = None
    static_x = torch.empty_like(xs[0])
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        static_l = fb(static_x)
    tiny = torch.zeros(4, device=DEV)
    bad = 0
    worst = [0.0, 0.0, 0.0]
    for i, x in enumerate(xs):
        static_x.copy_(x)
        graph.replay()
        if interleave:
            tiny.fill_(float(i))
            junk = torch.full((1 + i,), float("nan"), device=DEV)  # noqa: F841
        torch.cuda.synchronize()
        got = (static_l, w.grad, b.grad)
        ok = all(torch.equal(a, r) for a, r in zip(got, ref[i]))
        bad += 0 if ok else 1
        for k in range(3):
            worst[k] = max(worst[k], ((got[k].float() - ref[i][k].float()).abs().max() / ref[i][k].float().abs().max().clamp_min(1e-30)).item())
    print(f"{str(dtype)[6:]:8s} x[{T},{H}] w[{O},{H}] interleave={interleave}: {bad} of {steps} replayed steps differ; worst relative "
          f"difference loss {worst[0]:.2e} dW {worst[1]:.2e} db {worst[2]:.2e}")


if __name__ == "__main__":
    for dt in (torch.float32, torch.float16):
        for T, H, O in ((2048, 4096, 4096), (2048, 768, 768)):
            run(dt, T, H, O, interleave=True)
            run(dt, T, H, O, interleave=False)
