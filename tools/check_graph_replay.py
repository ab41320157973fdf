#!/usr/bin/env python3
"""Development check: do the bench's HIP graphs (this library's kernels only) keep producing the eager results
when small eager kernels are launched between replays?  (A graph of a whole PyTorch training step did not on
this ROCm build: with the optimizer's small eager kernels between replays it turned wrong after a few steps.)  Prints the number of steps whose packed
LWC gradient vector differs from the eager one."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    dev = "cuda:0"
    torch.cuda.set_device(0)
    wl = bench.Workload(dev, 2)
    fams = wl.families()
    for _n, fn in fams:
        fn()
    torch.cuda.synchronize()
    ref = wl.flat.clone()
    graphs = wl.capture()
    tiny = torch.zeros(4, device=dev)
    bad = 0
    for step in range(40):
        wl.flat.zero_()
        for _n, replay in graphs:
            replay()
            tiny.fill_(float(step))          # 1-block eager kernels between the replays
            tiny.mul_(1.0)
            _ = torch.ones((), device=dev)
        wl.flat.div_(1.0)
        torch.cuda.synchronize()
        if not torch.equal(wl.flat, ref):
            bad += 1
    print(f"steps with a gradient vector different from the eager run: {bad} of 40")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
