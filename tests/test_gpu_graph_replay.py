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


def test_capture_without_a_warm_statistics_table_is_refused():
    """The fresh-statistics activation quantizer keeps one statistics table per stream, zero-filled when it is created.
    Creating it inside a capture would turn the fill into a graph node and put the table into the graph's pool: the call
    must refuse, and after a warm-up on the capture stream the same capture must go through and replay the eager bits."""
    from miq import functional as F
    from miq import synth
    from miq._lib import MiqError
    dev = torch.device("cuda:0")
    x = synth.activation((256, 1024), 61).to(dev)
    layout = F.ClusterLayout.get([32] * 32, 1024, dev)
    ref = F.act_fresh_quant_fwd([x], 4, layout)[0][0].clone()
    side = torch.cuda.Stream(device=dev)          # a stream no call has run on yet: no table
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with pytest.raises(MiqError, match="capture"):
        with torch.cuda.graph(g, stream=side):
            F.act_fresh_quant_fwd([x], 4, layout)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):                 # warm-up on the capture stream creates the table eagerly
        F.act_fresh_quant_fwd([x], 4, layout)
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out = F.act_fresh_quant_fwd([x], 4, layout)[0][0]
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)


def test_cluster_layout_cannot_be_built_during_capture():
    """Its prefix sums are copied from pageable host memory, which a capturing stream refuses (and which invalidates the
    capture): a clear error instead, and the cached layout of an earlier eager call is fine."""
    from miq import functional as F
    from miq._lib import MiqError
    dev = torch.device("cuda:0")
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with pytest.raises(MiqError, match="capture"):
        with torch.cuda.graph(g, stream=side):
            F.ClusterLayout.get([3, 5, 8, 13, 21, 14], 64, dev)   # a layout no earlier test has used
    torch.cuda.synchronize()
    warm = F.ClusterLayout.get([3, 5, 8, 13, 21, 14], 64, dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        assert F.ClusterLayout.get([3, 5, 8, 13, 21, 14], 64, dev) is warm
    torch.cuda.synchronize()
