#!/usr/bin/env python3
"""Which PyTorch operations stop reproducing their eager results when replayed from a HIP graph on this ROCm build?
Each candidate is captured once on a static input and replayed on 8 different inputs."""
import torch

DEV = "cuda:0"


def check(name, fn, make, steps=8):
    xs = [make(i) for i in range(steps)]
    ref = [fn(x).clone() for x in xs]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(xs[0])
    torch.cuda.current_stream().wait_stream(side)
    static_x = torch.empty_like(xs[0])
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out = fn(static_x)
    bad, stale = 0, 0
    for i, x in enumerate(xs):
        static_x.copy_(x)
        g.replay()
        torch.cuda.synchronize()
        if not torch.equal(out, ref[i]):
            bad += 1
            if i > 0 and torch.equal(out, ref[i - 1]):
                stale += 1
    print(f"{name:44s} {bad} of {steps} replays differ ({stale} of them equal the PREVIOUS input's result)")


if __name__ == "__main__":
    gen = torch.Generator(device=DEV)
    big = lambda i: torch.randn(2048, 4096, device=DEV, generator=gen.manual_seed(i))
    small = lambda i: torch.randn(64, 64, device=DEV, generator=gen.manual_seed(i))
    check("sum of 8M elements (multi-block reduction)", lambda x: x.sum(), big)
    check("mean of squares of 8M elements", lambda x: (x * x).mean(), big)
    check("column sum [2048,4096] -> [4096]", lambda x: x.sum(0), big)
    check("row sum [2048,4096] -> [2048]", lambda x: x.sum(1), big)
    check("amax of 8M elements", lambda x: x.amax(), big)
    check("sum of 4096 elements (one block)", lambda x: x.sum(), small)
    check("elementwise x * 2 + 1", lambda x: x * 2 + 1, big)
    check("zeros_like + add", lambda x: torch.zeros_like(x) + x, big)
    check("layer_norm", lambda x: torch.nn.functional.layer_norm(x, (4096,)), big)
    check("softmax", lambda x: torch.softmax(x, -1), big)
    check("mse_loss against zeros", lambda x: torch.nn.functional.mse_loss(x, torch.zeros_like(x)), big)
    check("norm", lambda x: torch.norm(x), big)
