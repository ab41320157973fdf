#!/usr/bin/env python3
"""Do event-record nodes inside a captured HIP graph give usable timestamps on this build?  Captures the bench step as
ONE graph with `torch.cuda.Event(enable_timing=True, external=True)` recorded between the families, replays it and
prints the per-family times next to those of the one-graph-per-family form."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "llm-quantization_amd"))
import bench  # noqa: E402


def main():
    dev = "cuda:0"
    torch.cuda.set_device(0)
    wl = bench.Workload(dev, 0)
    fams = wl.families()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            for _n, fn in fams:
                fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    per = bench._capture_all(fams, side)
    # one graph per family
    seg = bench.Segments()
    for i in range(25):
        for name, fn in per:
            seg.run(name, fn, i >= 5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        for name, fn in per:
            fn()
    torch.cuda.synchronize()
    print("per-family graphs: step %.1f us" % ((time.perf_counter() - t0) / 20 * 1e6),
          {k: round(v / 20 * 1e3, 1) for k, v in seg.totals_ms().items()})
    # shared boundary events: 7 records per step instead of 12
    tot = {n: 0.0 for n, _ in per}
    wall = 0.0
    for i in range(25):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(per) + 1)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        evs[0].record()
        for j, (n, fn) in enumerate(per):
            fn()
            evs[j + 1].record()
        torch.cuda.synchronize()
        if i >= 5:
            wall += time.perf_counter() - t0
            for j, (n, _f) in enumerate(per):
                tot[n] += evs[j].elapsed_time(evs[j + 1])
    print("shared boundary events: step(incl sync) %.1f us" % (wall / 20 * 1e6), {k: round(v / 20 * 1e3, 1) for k, v in tot.items()},
          "sum %.1f" % (sum(tot.values()) / 20 * 1e3))
    # events around the last family only
    torch.cuda.synchronize()
    pairs = []
    t0 = time.perf_counter()
    for i in range(20):
        for n, fn in per:
            if n == "rowquant_bwd":
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                fn()
                b.record()
                pairs.append((a, b))
            else:
                fn()
    torch.cuda.synchronize()
    print("events around rowquant_bwd only: step %.1f us, family %.1f us" % ((time.perf_counter() - t0) / 20 * 1e6,
          sum(a.elapsed_time(b) for a, b in pairs) / 20 * 1e3))
    # one parent graph: child graph per family, event-record nodes between them (HIP runtime called directly)
    import ctypes
    path = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0]
    hip = ctypes.CDLL(path)
    vp = ctypes.c_void_p
    children = []
    for n, fn in fams:
        cg = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.graph(cg, stream=side):
            fn()
        children.append(cg)
    parent = vp()
    assert hip.hipGraphCreate(ctypes.byref(parent), 0) == 0
    evh = []
    for _ in range(len(fams) + 1):
        h = vp()
        assert hip.hipEventCreateWithFlags(ctypes.byref(h), 0) == 0
        evh.append(h)
    prev = vp()
    rc = hip.hipGraphAddEventRecordNode(ctypes.byref(prev), parent, None, ctypes.c_size_t(0), evh[0])
    assert rc == 0, ("event node", rc)
    for k, cg in enumerate(children):
        node = vp()
        dep = (vp * 1)(prev)
        rc = hip.hipGraphAddChildGraphNode(ctypes.byref(node), parent, dep, ctypes.c_size_t(1), vp(cg.raw_cuda_graph()))
        assert rc == 0, ("child node", rc)
        dep = (vp * 1)(node)
        prev = vp()
        rc = hip.hipGraphAddEventRecordNode(ctypes.byref(prev), parent, dep, ctypes.c_size_t(1), evh[k + 1])
        assert rc == 0, ("event node", rc)
    ex = vp()
    rc = hip.hipGraphInstantiate(ctypes.byref(ex), parent, None, None, ctypes.c_size_t(0))
    assert rc == 0, ("instantiate", rc)
    stream = vp(side.cuda_stream)
    tot = {n: 0.0 for n, _ in fams}
    for i in range(25):
        assert hip.hipGraphLaunch(ex, stream) == 0
        side.synchronize()
        if i >= 5:
            for j, (n, _f) in enumerate(fams):
                ms = ctypes.c_float()
                rc = hip.hipEventElapsedTime(ctypes.byref(ms), evh[j], evh[j + 1])
                assert rc == 0, ("elapsed", rc)
                tot[n] += ms.value
    side.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        hip.hipGraphLaunch(ex, stream)
    side.synchronize()
    print("one parent graph with event nodes: step %.1f us" % ((time.perf_counter() - t0) / 20 * 1e6),
          {k: round(v / 20 * 1e3, 1) for k, v in tot.items()}, "sum %.1f" % (sum(tot.values()) / 20 * 1e3))


if __name__ == "__main__":
    main()
