#!/usr/bin/env python3
"""A/B timing of the fused operator chains around the quantizer kernels in one process: the calibration step of
tools/bench_block.py with each class-wide switch turned off in turn (median of several runs).

  python tools/ab_block.py [--model 7b] [--amp] [--reps 5]
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_block  # noqa: E402
from miq import functional  # noqa: E402
from miq.blocks import _BlockBase, _Rotary  # noqa: E402
from miq.int_linear import QuantLinear  # noqa: E402
from miq.omni_norm import OmniLlamaRMSNorm  # noqa: E402

SWITCHES = {"fused_mse": (functional, "FUSED_MSE"), "half_weights": (_BlockBase, "half_weights"), "tensor_fold": (functional, "TENSOR_FOLD"), "rotary": (_Rotary, "fused"), "bias_grad": (QuantLinear, "fused_bias_grad"), "rmsnorm": (OmniLlamaRMSNorm, "fused")}


def one_run_ms(model, amp):
    return bench_block.run(model, 4, 2, amp, warmup=False)["s_per_step"] * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="7b")
    ap.add_argument("--amp", action="store_true")
    ap.add_argument("--reps", type=int, default=6)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    for _ in range(4):  # first-use costs, and the clocks settle under load
        bench_block.run(a.model, 4, 2, a.amp, warmup=False)
    # the configurations are interleaved (and their order rotated) so that clock drift hits all of them alike
    names = ["all_on"] + [f"{n}_off" for n in SWITCHES]
    times = {n: [] for n in names}
    for rep in range(a.reps):
        for k in range(len(names)):
            name = names[(k + rep) % len(names)]
            sw = SWITCHES.get(name[:-4]) if name != "all_on" else None
            if sw:
                setattr(sw[0], sw[1], False)
            times[name].append(one_run_ms(a.model, a.amp))
            if sw:
                setattr(sw[0], sw[1], True)
    out = {n: round(statistics.median(v), 3) for n, v in times.items()}
    print(json.dumps({"model": a.model, "amp": a.amp, "ms_per_step_median": out,
                      "all_on_runs": [round(v, 3) for v in times["all_on"]]}))


if __name__ == "__main__":
    main()
