#!/usr/bin/env python3
"""Which PyTorch operator (with which input shapes) launches which GPU kernels in one calibration step of a
decoder block: torch.profiler over the last epoch of tools/bench_block.py's run, grouped by operator + shapes and
sorted by GPU time.  Used to decide which operator chains around the quantizer kernels are worth fusing.

  python tools/trace_block_ops.py [--model 7b] [--amp] [--top 40]
"""
from __future__ import annotations

import argparse
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_block  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="7b")
    ap.add_argument("--amp", action="store_true")
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    bench_block.run(a.model, 2, 1, a.amp, warmup=False)  # first-use costs
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        bench_block.run(a.model, 2, 2, a.amp, warmup=False)
    print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=a.top,
                                                             max_name_column_width=60, max_shapes_column_width=90))


if __name__ == "__main__":
    main()
