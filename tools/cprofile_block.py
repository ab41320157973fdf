#!/usr/bin/env python3
"""Host-side profile (cProfile) of the calibration loop on the reference-log OPT-125m configuration, whose step is bound
by the host (small kernels): where the Python time of a step goes.

  python tools/cprofile_block.py [--samples 16] [--top 35]
"""
import argparse
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_block  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=16)
    ap.add_argument("--top", type=int, default=35)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    cfg = dict(bench_block.REFERENCE_LOG)
    cfg.pop("reference_s_per_layer")
    cfg["samples"] = a.samples
    bench_block.run(warmup=False, **cfg)
    pr = cProfile.Profile()
    pr.enable()
    out = bench_block.run(warmup=False, **cfg)
    pr.disable()
    print(out["s_per_step"])
    pstats.Stats(pr).sort_stats("tottime").print_stats(a.top)


if __name__ == "__main__":
    main()
