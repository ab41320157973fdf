#!/usr/bin/env python3
"""Time the activation families of the bench workload (forward calibrate + quantize, backward) per call group, from HIP
graphs over the bench's own tensors: median of 30 replays, microseconds per launch pair and fraction of 8 TB/s."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from miq import functional as F  # noqa: E402


def timed(fn, side, reps=30):
    with torch.cuda.stream(side):
        fn(); fn()
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        fn()
    ts = []
    with torch.cuda.stream(side):
        g.replay()
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); g.replay(); b.record()
            ts.append((a, b))
    side.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ts)
    return ms[len(ms) // 2] * 1e3


def main():
    dev = "cuda:0"
    wl = bench.Workload(dev, 2)
    side = torch.cuda.Stream()
    x2 = [x.reshape(-1, x.shape[-1]) for x in wl.x]
    g2 = [g.reshape(-1, g.shape[-1]) for g in wl.gx]
    tot_f = tot_b = 0.0
    nbytes = 0
    for ids in bench.SITE_GROUPS:
        xs, gs, lay = [x2[i] for i in ids], [g2[i] for i in ids], wl.layout[ids[0]]
        by = sum(x.numel() * x.element_size() for x in xs)
        saved = []

        def fwd():
            saved[:] = [F.act_fresh_quant_fwd(xs, bench.A_BITS, lay)[2]]

        def bwd():
            F.act_fresh_quant_bwd(gs, xs, saved[0], bench.A_BITS, lay)

        tf = timed(fwd, side)
        tb = timed(bwd, side)
        tot_f += tf; tot_b += tb; nbytes += by
        name = "+".join(bench.SITES[i][0] for i in ids)
        print(f"{name:24s} {str(xs[0].dtype)[6:]:8s} fwd {tf:7.2f} us {3 * by / tf / 1e3 / 8000:.3f}   bwd {tb:7.2f} us {3 * by / tb / 1e3 / 8000:.3f}")
    print(f"{'step':24s} {'':8s} fwd {tot_f:7.2f} us {3 * nbytes / tot_f / 1e3 / 8000:.3f}   bwd {tot_b:7.2f} us {3 * nbytes / tot_b / 1e3 / 8000:.3f}"
          f"   [{' '.join(k + '=' + v for k, v in os.environ.items() if k.startswith('MIQ_'))}]")


if __name__ == "__main__":
    main()
