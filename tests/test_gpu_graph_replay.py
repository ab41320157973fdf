"""The bench's HIP graphs (one per kernel family, this library's kernels only) must keep producing the eager results
when small eager kernels and allocations are interleaved between replays, which is what `bench.py --gpus N` does with its
all-reduce of the packed gradients: 12 steps, every family's graph followed by eager fills / a fresh allocation, the packed
LWC gradient vector and the activation outputs compared bit for bit with one eager pass."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_family_graphs_survive_interleaved_eager_work():
    sys.path.insert(0, REPO)
    import bench
    dev = "cuda:0"
    wl = bench.Workload(dev, 2)
    fams = wl.families()
    for _n, fn in fams:
        fn()
    torch.cuda.synchronize()
    ref = wl.flat.clone()
    assert torch.isfinite(ref).all() and ref.abs().sum() > 0
    graphs = wl.capture()
    tiny = torch.zeros(4, device=dev)
    junk = []
    for step in range(12):
        wl.flat.zero_()
        for _n, replay in graphs:
            replay()
            tiny.fill_(float(step))
            tiny.mul_(1.0)
            junk.append(torch.full((1 + step,), float("nan"), device=dev))  # fresh eager allocations, poisoned
            if len(junk) > 3:
                junk.pop(0)
        wl.flat.div_(1.0)
        torch.cuda.synchronize()
        assert torch.equal(wl.flat, ref), f"step {step}"
    del wl, graphs
    torch.cuda.empty_cache()
