#!/usr/bin/env python3
"""Eager against graph-replayed calibration steps on the bench's block configurations: per-step time and whether the loss
series are identical: python3 tools/dev/graph_step_ab.py"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "llm-quantization_amd"))
sys.path.insert(0, os.path.join(REPO, "tools"))
import bench_block as BB  # noqa: E402


def main():
    for name, kw in (("opt125m reference log", dict(BB.REFERENCE_LOG)),
                     ("7b amp 32x4", dict(model="7b", samples=32, epochs=4, amp=True)),
                     )[:int(os.environ.get("AB_CONFIGS", "2"))]:
        res = {}
        for graph in (False, True):
            res[graph] = BB.run(graph=graph, **kw)
        e, g = res[False], res[True]
        print(json.dumps({"config": name, "s_per_step_eager": e["s_per_step"], "s_per_step_graph": g["s_per_step"],
                          "s_layer_eager": e["s_layer_this_run"], "s_layer_graph": g["s_layer_this_run"],
                          "loss_last_eager": e["loss_last"], "loss_last_graph": g["loss_last"],
                          "same_last_loss": e["loss_last"] == g["loss_last"], "peak_GiB": [e["peak_mem_GiB"], g["peak_mem_GiB"]]}), flush=True)


if __name__ == "__main__":
    main()
