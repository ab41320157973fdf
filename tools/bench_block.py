#!/usr/bin/env python3
"""Block-calibration timing: one Llama-shaped decoder block through miq.calibrate.calibrate_layer
(full-precision targets, LET + LWC training steps with AdamW / GradScaler, in-place quantization and
the sequential propagation pass), synthetic weights and calibration samples.

Reports seconds per optimizer step and the projected seconds per layer for the reference's default
schedule (128 samples x 20 epochs, batch 1: omniquant.py:462-605).  The GEMMs / attention are
PyTorch-ROCm (rocBLAS / hipBLASLt); the quantizer ops are this repository's kernels.

  python tools/bench_block.py [--model 7b|13b|70b] [--samples 4] [--epochs 2] [--amp] [--act per_tensor|per_cluster|per_token]
With torch.distributed.run, every rank calibrates its own shard of the samples (data parallel,
gradient all-reduce per step).
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "llm-quantization_amd"))

from miq.blocks import QuantLlamaBlock, QuantOPTBlock, quant_args  # noqa: E402
from miq.calibrate import CalibConfig, calibrate_layer  # noqa: E402
from miq.dist import DPContext  # noqa: E402

SHAPES = {  # hidden, ffn, heads, kv heads
    "opt-125m": (768, 3072, 12, 12),
    "tiny": (256, 688, 4, 4),
    "7b": (4096, 11008, 32, 32),
    "13b": (5120, 13824, 40, 40),
    "70b": (8192, 28672, 64, 8),
}


# The one timing the reference publishes (log/opt-125m-w3a16/log_rank0_1698726878.txt:1,41): OPT-125m,
# exponent weight quantizer W3 symmetric, A8 static per_cluster without reorder, LET + LWC, 2 epochs,
# 128 samples x 2048 tokens, batch 1, fp32: 154.36 s for 12 layers on one unnamed NVIDIA GPU.
REFERENCE_LOG = {"model": "opt-125m", "samples": 128, "epochs": 2, "amp": False, "act": "per_cluster", "wbits": 3,
                 "abits": 8, "wexp": True, "wsym": True, "reference_s_per_layer": 154.36328053474426 / 12}


def synthetic_block(model, dev, args, seed=2):
    H, FF, heads, kv = SHAPES[model]
    g = torch.Generator(device=dev).manual_seed(seed)
    hd = H // heads

    def w(o, i):
        return torch.randn(o, i, generator=g, device=dev, dtype=torch.float16) * 0.02

    if model.startswith("opt"):
        def b(o):
            return torch.randn(o, generator=g, device=dev, dtype=torch.float16) * 0.02
        one = torch.ones(H, device=dev, dtype=torch.float16)
        weights = {"q": w(H, H), "k": w(H, H), "v": w(H, H), "o": w(H, H), "fc1": w(FF, H), "fc2": w(H, FF),
                   "q_b": b(H), "k_b": b(H), "v_b": b(H), "o_b": b(H), "fc1_b": b(FF), "fc2_b": b(H),
                   "ln1_w": one, "ln1_b": b(H), "ln2_w": one.clone(), "ln2_b": b(H)}
        return QuantOPTBlock(weights, heads, args).to(dev), (H, FF, heads, kv)

    weights = {"q": w(H, H), "k": w(kv * hd, H), "v": w(kv * hd, H), "o": w(H, H), "gate": w(FF, H), "up": w(FF, H),
               "down": w(H, FF), "ln1_w": torch.ones(H, device=dev, dtype=torch.float16),
               "ln2_w": torch.ones(H, device=dev, dtype=torch.float16)}
    return QuantLlamaBlock(weights, heads, args, num_kv_heads=kv, eps=1e-5).to(dev), (H, FF, heads, kv)


def run(model="7b", samples=4, epochs=2, amp=False, act="per_tensor", wbits=4, abits=4, seq=2048, dev="cuda:0",
        dp: DPContext | None = None, let=None, wexp=False, wsym=False, warmup=True, reference_s_per_layer=None, graph=False, fused_adamw=False):
    """Returns a dict of timings.  ``samples`` is the number of samples THIS rank holds."""
    dp = dp or DPContext()
    H, FF, heads, kv = SHAPES[model]
    if let is None:
        let = kv == heads  # the q/k smoothing of the reference assumes as many KV heads as heads
    is_llama = not model.startswith("opt")
    args = quant_args(wbits=wbits, abits=abits, lwc=True, a_dynamic_method=act, a_dynamic=(act == "per_token"),
                      metric="ema_minmax", let=let, weight_exp_quant=wexp, w_symmetric=wsym)
    if warmup:  # first-use costs (library handles, GEMM heuristics, allocator growth) stay out of the timing
        run(model, 2, 1, amp, act, wbits, abits, seq, dev, dp, let, wexp, wsym, warmup=False, graph=graph, fused_adamw=fused_adamw)
    qlayer, _ = synthetic_block(model, dev, args)
    g = torch.Generator(device=dev).manual_seed(100 + dp.rank)
    dtype = torch.float16 if amp else torch.float32
    inps = torch.randn(samples, seq, H, generator=g, device=dev, dtype=dtype)
    fp = inps.clone()
    mask = torch.full((seq, seq), torch.finfo(dtype).min, device=dev, dtype=dtype).triu(1).view(1, 1, seq, seq)
    cfg = CalibConfig(epochs=epochs, nsamples=samples * dp.world, batch_size=1, let=let, lwc=True, deactive_amp=not amp,
                      reorder="", graph_step=graph, fused_adamw=fused_adamw)
    prefix = "model.layers.0" if is_llama else "model.decoder.layers.0"
    sites = ("self_attn.q_proj", "self_attn.o_proj", "mlp.up_proj") if is_llama else ("self_attn.q_proj", "self_attn.out_proj", "fc1")
    act_scales = {f"{prefix}.{n}": torch.rand(H, generator=g, device=dev) * 3 + 0.2 for n in sites}
    act_shifts = {f"{prefix}.{n}": torch.randn(H, generator=g, device=dev) * 0.1 for n in sites}
    torch.cuda.reset_peak_memory_stats(dev)
    res = calibrate_layer(qlayer, inps, fp, mask, cfg, act_scales, act_shifts, prefix, dp=dp, is_llama=is_llama)
    sec = res["seconds"]
    steps = samples * epochs
    s_step = sec["train"] / max(steps, 1)
    out = {
        "model": (f"Llama-2-{model}" if is_llama else model) + " block", "seq_len": seq, "samples_per_rank": samples, "epochs": epochs, "ranks": dp.world,
        "w_bits": wbits, "a_bits": abits, "weight_quantizer": "exp-v2" if wexp else "uniform", "w_symmetric": wsym,
        "act_quant": act, "let": let, "precision": "fp16 autocast" if amp else "fp32",
        # what RAN, not what was asked for: a step that could not be captured or did not reproduce the eager step falls back
        "step": ("HIP graph replay (forward + loss + backward), optimizer eager" if res["graph_steps"] > 0 else
                 ("eager (graph replay was requested and given up: see notes)" if graph else "eager")),
        "graph_steps": res["graph_steps"], "steps": res["steps"], "notes": res["notes"],
        "s_per_step": round(s_step, 5), "s_fp_targets_per_sample": round(sec["fp_targets"] / samples, 5),
        "s_propagate_per_sample": round(sec["propagate"] / (samples * dp.world), 5),
        "loss_first": res["losses"][0], "loss_last": res["losses"][-1],
        "peak_mem_GiB": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
    }
    # the reference schedule: 128 samples, 20 epochs; samples are split over the ranks, the propagation
    # pass is replicated
    n, e = 128, 20
    out["s_per_layer_projected_128x20"] = round((n / dp.world) * e * s_step + (n / dp.world) * out["s_fp_targets_per_sample"]
                                                + n * out["s_propagate_per_sample"], 2)
    out["s_layer_this_run"] = round(sum(sec.values()), 3)
    if reference_s_per_layer:
        out["reference_s_per_layer"] = round(reference_s_per_layer, 3)
        out["speedup_vs_reference_log"] = round(reference_s_per_layer / out["s_layer_this_run"], 2)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="7b", choices=sorted(SHAPES))
    ap.add_argument("--samples", type=int, default=4)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--amp", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay forward + loss + backward of a step from one HIP graph")
    ap.add_argument("--fused-adamw", action="store_true",
                    help="torch's fused AdamW (no host synchronisation in GradScaler.step; its update differs in the last bits, "
                         "which moves the trajectory: not the reference's arithmetic)")
    ap.add_argument("--act", default="per_tensor", choices=["per_tensor", "per_cluster", "per_token"])
    ap.add_argument("--abits", type=int, default=4)
    ap.add_argument("--wbits", type=int, default=4)
    ap.add_argument("--wexp", action="store_true", help="exponent (v2) weight quantizer")
    ap.add_argument("--wsym", action="store_true")
    ap.add_argument("--reference-log", action="store_true", help="the configuration of the reference's committed run log")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dp = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        dp = DPContext.from_env()
    if a.reference_log:
        out = run(seq=a.seq, dev=f"cuda:{local}", dp=dp, graph=a.graph, fused_adamw=a.fused_adamw, **REFERENCE_LOG)
    else:
        out = run(a.model, a.samples, a.epochs, a.amp, a.act, a.wbits, a.abits, a.seq, f"cuda:{local}", dp, wexp=a.wexp,
                  wsym=a.wsym, graph=a.graph, fused_adamw=a.fused_adamw)
    if int(os.environ.get("RANK", "0")) == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
