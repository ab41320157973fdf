#!/usr/bin/env python3
"""Headline benchmark: the fake-quantization hot path of one OmniQuant calibration step on a
Llama-2-7B-shaped block (the configuration the headline metric is quoted on: W4A4, static per-cluster activation
quantization with RPTQ channel reorder), synthetic data, one MI355X per rank.

One "step" = everything the quantizer kernels do for ONE calibration sample of 2048 tokens
(GEMMs / attention / layer norms are not part of this path and are not executed):
  forward   7 weight matrices row-wise W4 fake-quant with learnable clipping (fp32), batched per row width
            (6 matrices [*, 4096] in one launch, [4096, 11008] in another)
            2 reorder gathers of the residual stream (fp16, [1,2048,4096]), both in one launch
            10 activation sites: per-cluster EMA min/max calibration + A4 quantization; sites whose inputs exist side
            by side in a decoder block go through one call (the three q/k/v inputs, the three q/k/v projection
            outputs, the gate/up inputs: 5 calls of 2 launches for the 10 sites, forward and backward)
              q/k/v inputs   3 x [1,2048,4096] fp32 (layer-norm output), 32 clusters x 128 channels
              q, k, v, o-in  4 x [1,2048,4096] fp16, 128 clusters x 32 channels (4 per head)
              gate/up inputs 2 x [2048,4096]   fp32, 32 clusters x 128
              down input     1 x [2048,11008]  fp16, 32 clusters x 344
  backward  straight-through gradients of all of the above (input gradients included, as with
            LET enabled) and, with N > 1 ranks, one RCCL all-reduce of the packed LWC gradients
            (84,992 floats) — the data-parallel exchange of the calibration loop.

value = algorithmic bytes of the step (read-once/write-once accounting: fwd 2ns, bwd 3ns, calibrate+
quantize 3ns, gather 2ns) summed over all ranks / wall time of the slowest rank.  Inputs are resident
in HBM before the timed region.

Usage: python bench.py [--gpus N] [--steps K] [--warmup W]
With N > 1 and no WORLD_SIZE in the environment the script starts its own ranks (one per GPU) through
torch.distributed.run as a child process and relays rank 0's line; under torch.distributed.run it is a rank.
Prints ONE JSON line on rank 0.

Beside the timed step the line carries, measured outside the timed region: `families_c3_c4` (the exponent and mixed-bit
quantizer families on the Llama-2-7B / Llama-2-13B weight shapes: the other two of the five operator families),
`block_calib` (seconds per layer of the calibration loop) and `cpu_baseline`.

--workload c4 runs the row-sharded weight workload instead: the seven matrices of one Llama-2-13B layer (fp32) through the
learnable mixed-bit quantizer in its soft and hard phases (forward + backward including d n_bits) and the fixed mixed-bit
quantizer (forward + backward), every matrix sharded by output channel over the ranks (no collective: rows are
independent), then one all-gather per matrix of the fixed mixed-bit dequantized shards so that every rank holds the whole
layer.  The total work is fixed, so this is a strong-scaling measurement.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "llm-quantization_amd"))
sys.path.insert(0, REPO)

from miq import functional as F  # noqa: E402
from miq import synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
T = 2048
H, FF, HEADS = 4096, 11008, 32
WEIGHTS = [("q", H, H), ("k", H, H), ("v", H, H), ("o", H, H), ("gate", FF, H), ("up", FF, H), ("down", H, FF)]
F32, F16 = torch.float32, torch.float16
SITES = [  # name, shape, dtype, cluster sizes
    ("q_in", (1, T, H), F32, [H // 32] * 32), ("k_in", (1, T, H), F32, [H // 32] * 32),
    ("v_in", (1, T, H), F32, [H // 32] * 32),
    ("q_out", (1, T, H), F16, [H // HEADS // 4] * (4 * HEADS)), ("k_out", (1, T, H), F16, [H // HEADS // 4] * (4 * HEADS)),
    ("v_out", (1, T, H), F16, [H // HEADS // 4] * (4 * HEADS)), ("o_in", (1, T, H), F16, [H // HEADS // 4] * (4 * HEADS)),
    ("gate_in", (T, H), F32, [H // 32] * 32), ("up_in", (T, H), F32, [H // 32] * 32),
    ("down_in", (T, FF), F16, [FF // 32] * 32),
]
W_BITS, A_BITS = 4, 4
# sites quantized by one call: their inputs are available together in a block (models/int_llama_layer.py:110-116 calls the
# q / k / v input quantizers on one tensor; the projection outputs feed the two quantized matmuls, :139-154)
SITE_GROUPS = [[0, 1, 2], [3, 4, 5], [6], [7, 8], [9]]
# the rocprofv3 kernel behind every family (the activation families are chains; their streaming kernel is named)
FAMILY_KERNEL = {"rowquant_fwd": "k_uniform_fwd_batch", "rowquant_bwd": "k_uniform_bwd_batch",
                 "act_calib_quant_fwd": "k_colstat_slots + k_colquant_slots", "act_bwd": "k_colbwd_stream + k_colbwd_pairs",
                 "gather_fwd": "k_gather", "gather_bwd": "k_gather"}


class Segments:
    """HIP-event brackets around kernel families, recorded on the stream the kernels run on.  An event record is not free
    on this GPU: about 5 us of stream time each (a step of six families bracketed one by one ran 1.14 ms, the same six
    graphs without events 1.06 ms; an event-record NODE inside one graph of the whole step costs the same, and torch
    refuses external events on ROCm anyway).  So the timed region brackets only the dominant family (`only`), and the
    table of all families comes from an instrumented pass of the same steps in front of it (`boundaries`: one event at
    each family boundary, shared by the two neighbours)."""

    def __init__(self, only=None):
        self.ev = {}
        self.only = only

    def run(self, name, fn, timed):
        if not timed or (self.only is not None and name not in self.only):
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        self.ev.setdefault(name, []).append((a, b))
        return r

    def boundaries(self, fams):
        """One pass over `fams` with an event at every boundary."""
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(len(fams) + 1)]
        marks[0].record()
        for i, (name, fn) in enumerate(fams):
            fn()
            marks[i + 1].record()
            self.ev.setdefault(name, []).append((marks[i], marks[i + 1]))

    def totals_ms(self):
        return {k: sum(a.elapsed_time(b) for a, b in v) for k, v in self.ev.items()}


def _capture_all(fams, side):
    """One HIP graph per family, captured on the stream the warm-up ran on (per-stream state of the library -- the
    statistics table of the activation kernels -- is warm there, so nothing is allocated inside a capture)."""
    pool = torch.cuda.graph_pool_handle()
    out = []
    for name, fn in fams:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=pool, stream=side):
            fn()
        out.append((name, g.replay))
    return out


class Workload:
    def __init__(self, dev, seed):
        self.dev = dev
        self.w, self.gw, self.up, self.low = [], [], [], []
        for i, (_n, o, c) in enumerate(WEIGHTS):
            self.w.append(synth.weight(o, c, seed + i).to(dev))
            self.gw.append((synth.grad_like(self.w[-1].cpu(), seed + 50 + i) * 1e-3).to(dev))
            u, l = synth.lwc_clips(o, seed + 100 + i)
            self.up.append(u.to(dev))
            self.low.append(l.to(dev))
        self.x, self.gx, self.layout = [], [], []
        for i, (_n, shape, dt, counts) in enumerate(SITES):
            a = synth.activation(shape, seed + 200 + i, dt)
            self.x.append(a.to(dev))
            self.gx.append((synth.grad_like(a, seed + 300 + i) * 1e-3).to(dev))
            self.layout.append(F.ClusterLayout.get(counts, shape[-1], torch.device(dev)))
        self.res = [synth.activation((1, T, H), seed + 400 + i, F16).to(dev) for i in range(2)]
        self.gres = [torch.randn_like(r) for r in self.res]
        self.idx = [synth.permutation(H, seed + 500 + i).to(torch.int32).to(dev) for i in range(2)]
        self.inv = [torch.argsort(i.long()).to(torch.int32) for i in self.idx]
        self.n_lwc = sum(2 * o for _n, o, _c in WEIGHTS)
        self.flat = torch.empty(self.n_lwc, dtype=torch.float32, device=dev)

    # algorithmic bytes per family and launches per step
    def bytes(self):
        wb = sum(o * c * 4 for _n, o, c in WEIGHTS)
        ab = sum(x.numel() * x.element_size() for x in self.x)
        gb = sum(r.numel() * r.element_size() for r in self.res)
        return {"rowquant_fwd": 2 * wb, "rowquant_bwd": 3 * wb, "act_calib_quant_fwd": 3 * ab, "act_bwd": 3 * ab,
                "gather_fwd": 2 * gb, "gather_bwd": 2 * gb}

    def launches(self):
        n_w = len({c for _n, _o, c in WEIGHTS})  # one launch per row width
        return {"rowquant_fwd": n_w, "rowquant_bwd": n_w, "act_calib_quant_fwd": 2 * len(SITE_GROUPS),
                "act_bwd": 2 * len(SITE_GROUPS), "gather_fwd": 1, "gather_bwd": 1}

    def families(self):
        """Ordered (name, callable) pairs; every callable only enqueues kernels on the current stream."""
        saved, grads = [], []

        # weight matrices of one row width go through one launch (6 x [*, 4096] and 1 x [4096, 11008])
        groups = {}
        for i, (_n, _o, c) in enumerate(WEIGHTS):
            groups.setdefault(c, []).append(i)
        self.w_groups = list(groups.values())

        def w_fwd():
            for ids in self.w_groups:
                F.rowquant_uniform_fwd_batch([self.w[i] for i in ids], W_BITS, False, [self.up[i] for i in ids],
                                             [self.low[i] for i in ids])

        def g_fwd():  # the inputs of the block's two reordered norms, one launch
            F.gather_lastdim_batch(self.res, self.idx)

        x2 = [x.reshape(-1, x.shape[-1]) for x in self.x]
        g2 = [g.reshape(-1, g.shape[-1]) for g in self.gx]

        def a_fwd():
            saved.clear()
            for ids in SITE_GROUPS:
                _outs, _codes, vecs = F.act_fresh_quant_fwd([x2[i] for i in ids], A_BITS, self.layout[ids[0]])
                saved.append(vecs)

        def a_bwd():
            for ids, vecs in zip(SITE_GROUPS, saved):
                F.act_fresh_quant_bwd([g2[i] for i in ids], [x2[i] for i in ids], vecs, A_BITS, self.layout[ids[0]])

        def g_bwd():
            F.gather_lastdim_batch(self.gres, self.inv)

        def w_bwd():
            grads.clear()
            for ids in self.w_groups:
                _dx, dus, dls = F.rowquant_uniform_bwd_batch([self.gw[i] for i in ids], [self.w[i] for i in ids], W_BITS, False,
                                                             [self.up[i] for i in ids], [self.low[i] for i in ids])
                for du, dl in zip(dus, dls):
                    grads.extend((du, dl))
            torch.cat([g.reshape(-1) for g in grads], out=self.flat)

        return [("rowquant_fwd", w_fwd), ("gather_fwd", g_fwd), ("act_calib_quant_fwd", a_fwd), ("act_bwd", a_bwd),
                ("gather_bwd", g_bwd), ("rowquant_bwd", w_bwd)]

    def capture(self):
        """Capture every family into a HIP graph (removes the Python/launch overhead of ~90 small
        launches per step; the kernels neither allocate nor synchronise, so they are capturable)."""
        fams = self.families()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                for _n, fn in fams:
                    fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        return _capture_all(fams, side)

    def step(self, fams, seg: Segments, timed: bool, world: int, boundaries: bool = False):
        if boundaries:
            seg.boundaries(fams)
        else:
            for name, fn in fams:
                seg.run(name, fn, timed)
        if world > 1:
            if os.environ.get("MIQ_BENCH_ONE_GPU") == "1":  # rehearsal: gloo reduces a host copy
                host = self.flat.cpu()
                torch.distributed.all_reduce(host)
                self.flat.copy_(host)
            else:
                torch.distributed.all_reduce(self.flat)
            self.flat.div_(world)


L13 = [("q", 5120, 5120), ("k", 5120, 5120), ("v", 5120, 5120), ("o", 5120, 5120), ("gate", 13824, 5120), ("up", 13824, 5120),
       ("down", 5120, 13824)]
C4_KERNEL = {"mix_soft_fwd": "k_mix_fwd_batch", "mix_soft_bwd": "k_mix_bwd_batch", "mix_hard_fwd": "k_mix_fwd_batch",
             "mix_hard_bwd": "k_mix_bwd_batch",
             "mix2_fwd": "k_uniform_fwd_bits_batch", "mix2_bwd": "k_uniform_bwd_bits_batch",
             "all_gather": "RCCL all-gather (not a kernel of this library)"}


class C4Workload:
    """Row shards of one Llama-2-13B layer on this rank.  Every rank draws the same full matrices (same seed), keeps
    its rows (miq.dist.shard_weight_rows) and drops the rest."""

    def __init__(self, dev, ctx, rehearsal):
        from miq.dist import shard_weight_rows
        self.dev, self.ctx, self.rehearsal = dev, ctx, rehearsal
        self.rows_total = [o for _n, o, _c in L13]
        self.w, self.g, self.up, self.low, self.bits, self.fixed = [], [], [], [], [], []
        gen = torch.Generator(device=dev)
        for i, (_n, o, c) in enumerate(L13):
            gen.manual_seed(1300 + i)
            full = torch.randn(o, c, device=dev, generator=gen) * 0.02
            (sl, shard), = shard_weight_rows([full], ctx)
            self.w.append(shard.clone())
            del full
            rows = self.w[-1].shape[0]
            self.g.append(torch.randn(rows, c, device=dev, generator=gen) * 1e-3)
            u, l = synth.lwc_clips(o, 1400 + i)
            self.up.append(u[sl].contiguous().to(dev))
            self.low.append(l[sl].contiguous().to(dev))
            self.bits.append(synth.mix_bits(o, 3.0, 1500 + i)[sl].contiguous().to(dev))
            self.fixed.append(synth.mix2_bits(o, 1600 + i)[sl].contiguous().to(dev))
        self.full = [torch.empty(o, c, device=dev) for _n, o, c in L13] if ctx.active else None
        self.deq = [None] * len(L13)

    def bytes(self):
        """Algorithmic bytes of the WHOLE layer per step (the shards of all ranks together)."""
        n = sum(o * c * 4 for _n, o, c in L13)
        by = {"mix_soft_fwd": 2 * n, "mix_soft_bwd": 3 * n, "mix_hard_fwd": 2 * n, "mix_hard_bwd": 3 * n, "mix2_fwd": 2 * n,
              "mix2_bwd": 3 * n}
        return by

    def launches(self, family):
        """Launches per step: one per row width in every family."""
        return len({c for _n, _o, c in L13})

    def gather_bytes_in(self):
        """Bytes that arrive at every rank in the all-gather of one step."""
        w = self.ctx.world
        return sum(o * c * 4 for _n, o, c in L13) * (w - 1) // w

    def families(self):
        # matrices of one row width go through one launch each way (6 x [*, 5120] and 1 x [5120, 13824])
        by_width = {}
        for i, (_n, _o, c) in enumerate(L13):
            by_width.setdefault(c, []).append(i)
        self.w_groups = list(by_width.values())

        def mix(hard, bwd):
            def run():
                for grp in self.w_groups:
                    sel = lambda ts: [ts[i] for i in grp]
                    if bwd:
                        F.rowquant_mix_bwd_batch(sel(self.g), sel(self.w), sel(self.bits), False, sel(self.up), sel(self.low),
                                                 hard=hard, want_dx=True)
                    else:
                        F.rowquant_mix_fwd_batch(sel(self.w), sel(self.bits), False, sel(self.up), sel(self.low), hard=hard)
            return run

        def mix2_fwd():
            for grp in self.w_groups:
                sel = lambda ts: [ts[i] for i in grp]
                outs = F.rowquant_mix2_fwd_batch(sel(self.w), sel(self.fixed), False, sel(self.up), sel(self.low))[0]
                for i, o in zip(grp, outs):
                    self.deq[i] = o

        def mix2_bwd():
            for grp in self.w_groups:
                sel = lambda ts: [ts[i] for i in grp]
                F.rowquant_mix2_bwd_batch(sel(self.g), sel(self.w), sel(self.fixed), False, sel(self.up), sel(self.low), want_dx=True)

        return [("mix_soft_fwd", mix(False, False)), ("mix_soft_bwd", mix(False, True)), ("mix_hard_fwd", mix(True, False)),
                ("mix_hard_bwd", mix(True, True)), ("mix2_fwd", mix2_fwd), ("mix2_bwd", mix2_bwd)]

    def capture(self):
        fams = self.families()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                for _n, fn in fams:
                    fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        return _capture_all(fams, side)

    def all_gather(self):
        from miq.dist import gather_row_shards
        for i, rows in enumerate(self.rows_total):
            if self.rehearsal:  # gloo moves host copies
                host = gather_row_shards(self.deq[i].cpu(), rows, self.ctx)
                self.full[i].copy_(host)
            else:
                gather_row_shards(self.deq[i], rows, self.ctx, out=self.full[i])

    def step(self, fams, seg, timed):
        for name, fn in fams:
            seg.run(name, fn, timed)
        if self.ctx.active:
            seg.run("all_gather", self.all_gather, timed)


def run_c4(args, world, rank, dev, rehearsal, barrier):
    from miq.dist import DPContext
    ctx = DPContext.from_env()
    wl = C4Workload(dev, ctx, rehearsal)
    fams = wl.families() if args.eager else wl.capture()
    idle = Segments()
    for _ in range(args.warmup):
        wl.step(fams, idle, False)
    # every family (and the all-gather) bracketed, outside the timed region; the timed region brackets the dominant one
    seg_all = Segments()
    for _ in range(args.steps):
        wl.step(fams, seg_all, True)
    torch.cuda.synchronize()
    totals_all = seg_all.totals_ms()
    dom = max(wl.bytes(), key=lambda k: totals_all[k])
    seg = Segments(only={dom})
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step(fams, seg, True)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        return
    totals = seg.totals_ms()  # the timed region: the dominant family only
    by = wl.bytes()
    step_bytes = sum(by.values())
    # rank 0's own kernels move 1/world of the layer: per-rank GB/s against this GPU's HBM peak
    fam = {k: {"kernel": C4_KERNEL[k], "GBps_per_rank": round(by[k] / world * args.steps / (totals_all[k] * 1e-3) / 1e9, 1),
               "frac_hbm_peak": round(by[k] / world * args.steps / (totals_all[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               "ms_per_step": round(totals_all[k] / args.steps, 4), "launches_per_step": wl.launches(k)} for k in by}
    if world > 1:
        ms = totals_all["all_gather"] / args.steps
        fam["all_gather"] = {"kernel": C4_KERNEL["all_gather"], "ms_per_step": round(ms, 4),
                             "GBps_into_each_rank": round(wl.gather_bytes_in() / (ms * 1e-3) / 1e9, 1),
                             "collectives_per_step": len(L13),
                             "transport": "gloo over host copies (one-GPU rehearsal)" if rehearsal else "RCCL over xGMI"}
    achieved = by[dom] / world * args.steps / (totals[dom] * 1e-3) / 1e9
    out = {
        "metric": "fake-quant GB/s (% HBM peak) + OmniQuant block-calib s/layer, Llama-2-7B",
        "value": round(step_bytes * args.steps / dt / 1e9, 1),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "Llama-2-13B layer, mixed-bit weights with per-output-channel learnable bits: learnable mixed-bit "
                               "soft + hard phases fwd+bwd (incl. d n_bits) and fixed mixed-bit (10/80/10 % of 4/3/2 bit) fwd+bwd "
                               "on the 7 matrices (fp32), rows sharded over the ranks"
                               + (", all-gather of the fixed mixed-bit dequantized shards (inside the timed region)" if world > 1 else ""),
                   "rows_per_rank": [wl.w[i].shape[0] for i in range(len(L13))],
                   "launch": "eager" if args.eager else "hipGraph replay per kernel family",
                   "parallelism": f"row shards x{world}" + (" + all-gather" if world > 1 else ""),
                   "note": "value = bytes of the whole layer / wall time of the slowest rank, all-gather included; no hardware "
                           "scaling curve exists for this workload until it is run on a multi-GPU node"},
        "frac_hbm_peak": round(step_bytes / world * args.steps / dt / 1e9 / HBM_PEAK_GBS, 4),
        "roofline": {"bound": "hbm", "kernel": C4_KERNEL[dom], "family": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "traffic_source": None,
                     "bytes_per_launch": by[dom] // world // wl.launches(dom),
                     "avg_launch_us": round(totals[dom] * 1e3 / (wl.launches(dom) * args.steps), 2)},
        "families": fam,
        "families_timing": f"instrumented pass of {args.steps} steps in front of the timed region (HIP events around every family, "
                           f"about 5 us of stream time per event record); the timed region brackets only {dom}",
    }
    print(json.dumps(out))


L7 = [(o, c) for _n, o, c in WEIGHTS]
C34_KERNEL = {"exp2": ("k_exp2_fwd_batch", "k_exp2_bwd_batch"), "exp1": ("k_exp1_fwd_batch", "k_exp1_bwd_batch"),
              "mix_soft": ("k_mix_fwd_batch", "k_mix_bwd_batch"), "mix_hard": ("k_mix_fwd_batch", "k_mix_bwd_batch"),
              "mix2": ("k_uniform_fwd_bits_batch", "k_uniform_bwd_bits_batch")}


def families_c3_c4(dev, replays=20, rounds=3):
    """The exponent (v2 and v1, W3, Llama-2-7B matrices) and mixed-bit (learnable soft / hard, fixed; Llama-2-13B
    matrices) weight quantizers, forward and backward with input gradient, fp32 with learnable clipping: the operator
    families the timed step does not contain.  One HIP graph per family and direction over the seven matrices of a
    layer (0.8 / 1.3 GB per operand: several times the 256 MiB Infinity Cache, so back-to-back replays find nothing
    cached), timed with one pair of HIP events around `replays` replays, `rounds` times; the median round gives GB/s of
    algorithmic bytes (forward 2ns, backward 3ns)."""
    out = {}
    side = torch.cuda.Stream(device=dev)

    def timed(fn):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
            fn()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            fn()
        ts = []
        with torch.cuda.stream(side):
            g.replay()
            for _ in range(rounds):  # `replays` replays between one pair of events: the event records' own time is spread thin
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(replays):
                    g.replay()
                b.record()
                ts.append((a, b))
        side.synchronize()
        ms = sorted(a.elapsed_time(b) / replays for a, b in ts)
        return ms[len(ms) // 2] * 1e-3

    def layer(shapes, seed):
        gen = torch.Generator(device=dev).manual_seed(seed)
        xs = [torch.randn(o, c, device=dev, generator=gen) * 0.02 for o, c in shapes]
        gs = [torch.randn(o, c, device=dev, generator=gen) * 1e-3 for o, c in shapes]
        clips = [synth.lwc_clips(o, seed + 10 + i) for i, (o, _c) in enumerate(shapes)]
        ups = [u.to(dev) for u, _l in clips]
        lows = [l.to(dev) for _u, l in clips]
        widths = {}
        for i, (_o, c) in enumerate(shapes):
            widths.setdefault(c, []).append(i)
        return xs, gs, ups, lows, list(widths.values()), sum(o * c for o, c in shapes) * 4

    def record(name, shapes_tag, fwd, bwd, nbytes, launches):
        for tag, fn, mult, kern in (("fwd", fwd, 2, C34_KERNEL[name][0]), ("bwd", bwd, 3, C34_KERNEL[name][1])):
            t = timed(fn)
            gbs = mult * nbytes / t / 1e9
            out[f"{name}_{tag}"] = {"kernel": kern, "shapes": shapes_tag, "GBps": round(gbs, 1),
                                    "frac_hbm_peak": round(gbs / HBM_PEAK_GBS, 4), "launches": launches,
                                    "us_per_launch": round(t / launches * 1e6, 2), "replays": replays, "rounds": rounds}

    # exponent quantizers, W3, the seven Llama-2-7B matrices
    xs, gs, ups, lows, groups, nbytes = layer(L7, 700)
    sel = lambda ts, ids: [ts[i] for i in ids]
    record("exp2", "Llama-2-7B layer (7 matrices), W3, fp32, LWC",
           lambda: [F.rowquant_exp2_fwd_batch(sel(xs, g), 3, sel(ups, g), sel(lows, g)) for g in groups],
           lambda: [F.rowquant_exp2_bwd_batch(sel(gs, g), sel(xs, g), 3, sel(ups, g), sel(lows, g)) for g in groups],
           nbytes, len(groups))
    record("exp1", "Llama-2-7B layer (7 matrices), W3 symmetric, fp32, LWC",
           lambda: [F.rowquant_exp1_fwd_batch(sel(xs, g), 3, True, sel(ups, g), sel(lows, g)) for g in groups],
           lambda: [F.rowquant_exp1_bwd_batch(sel(gs, g), sel(xs, g), 3, True, sel(ups, g), sel(lows, g)) for g in groups],
           nbytes, len(groups))
    del xs, gs, ups, lows
    torch.cuda.empty_cache()
    # mixed-bit quantizers, the seven Llama-2-13B matrices
    shapes13 = [(o, c) for _n, o, c in L13]
    xs, gs, ups, lows, groups, nbytes = layer(shapes13, 1300)
    soft = [synth.mix_bits(o, 3.0, 1500 + i).reshape(-1).contiguous().to(dev) for i, (o, _c) in enumerate(shapes13)]
    hard = [b.round() for b in soft]
    fixed = [synth.mix2_bits(o, 1600 + i).contiguous().to(dev) for i, (o, _c) in enumerate(shapes13)]
    for name, bits, is_hard in (("mix_soft", soft, False), ("mix_hard", hard, True)):
        record(name, "Llama-2-13B layer (7 matrices), n_bits = 3.5 + U(0, 0.25)" + (" rounded" if is_hard else "") + ", fp32, LWC",
               lambda: [F.rowquant_mix_fwd_batch(sel(xs, g), sel(bits, g), False, sel(ups, g), sel(lows, g), hard=is_hard)
                        for g in groups],
               lambda: [F.rowquant_mix_bwd_batch(sel(gs, g), sel(xs, g), sel(bits, g), False, sel(ups, g), sel(lows, g),
                                                 hard=is_hard, want_dx=True) for g in groups],
               nbytes, len(groups))
    record("mix2", "Llama-2-13B layer (7 matrices), 10/80/10 % of rows at 4/3/2 bits, fp32, LWC",
           lambda: [F.rowquant_mix2_fwd_batch(sel(xs, g), sel(fixed, g), False, sel(ups, g), sel(lows, g)) for g in groups],
           lambda: [F.rowquant_mix2_bwd_batch(sel(gs, g), sel(xs, g), sel(fixed, g), False, sel(ups, g), sel(lows, g),
                                              want_dx=True) for g in groups],
           nbytes, len(groups))
    del xs, gs, ups, lows
    torch.cuda.empty_cache()
    return out


def cpu_baseline(threads: int):
    """The CPU oracle (bit-checked against the reference's golden vectors) on a bounded sample of the
    same workload: one [4096,4096] weight forward+backward and one [1,2048,4096] fp32 activation site
    forward+backward, on the host cores of this box."""
    from oracle import miq_oracle as O

    torch.set_num_threads(threads)
    w = synth.weight(H, H, 2)
    up, low = synth.lwc_factors(H, 3)  # raw factors: the oracle applies the sigmoid itself
    gw = synth.grad_like(w, 4)
    a = synth.activation((1, T, H), 5)
    ga = synth.grad_like(a, 6)
    counts = [H // 32] * 32

    def once():
        wr = w.clone().requires_grad_(True)
        u, l = up.clone().requires_grad_(True), low.clone().requires_grad_(True)
        r = O.uniform_row(wr, W_BITS, False, u, l)
        torch.autograd.grad((r.out * gw).sum(), [wr, u, l])
        ar = a.clone().requires_grad_(True)
        out, _ = O.act_forward(ar, O.ActState(), A_BITS, "calibration", method="per_cluster", metric="ema_minmax",
                               cluster_dim=2, cluster_counts=counts)
        torch.autograd.grad((out * ga).sum(), [ar])

    once()
    t0 = time.perf_counter()
    n = 0
    while True:
        once()
        n += 1
        dt = time.perf_counter() - t0
        if dt > 12.0 or n >= 100:
            break
    nbytes = (2 + 3) * w.numel() * 4 + (3 + 3) * a.numel() * 4
    return {"value": round(nbytes * n / dt / 1e9, 3), "unit": "GB/s", "cores": threads, "kind": "port",
            "sample": f"{n} x (uniform W4+LWC fwd+bwd on [4096,4096] fp32 + per-cluster A4 calibrate+quantize "
                      f"fwd+bwd on [1,2048,4096] fp32), torch CPU oracle, {dt:.1f} s"}


def block_calib(dev, world, rank, rehearsal=False):
    """Second half of the metric: seconds per layer of the block-calibration loop (GEMMs and attention
    are PyTorch-ROCm, the quantizer ops are the kernels benchmarked above).  Outside the timed region
    of `value`.  One rank: the reference's own published configuration (its committed OPT-125m run
    log) and a Llama-2-7B block in the fp32 and fp16-autocast paths; N ranks: the Llama-2-7B block with
    the samples sharded over the ranks and the per-step gradient all-reduce."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import bench_block as BB
    from miq.dist import DPContext

    def say(msg):  # progress on stderr: this part runs for a minute without a result line
        if rank == 0:
            print(f"[bench] block calibration: {msg}", file=sys.stderr, flush=True)

    out = {}
    try:
        if world == 1:
            def graph_or_eager(**kw):
                """The step replayed from one HIP graph (same trajectory, checked against eager steps while it runs); the
                eager loop if that check ever objects."""
                try:
                    return BB.run(dev=dev, graph=True, **kw)
                except RuntimeError as e:
                    r = BB.run(dev=dev, graph=False, **kw)
                    r["graph_step_refused"] = str(e)[:200]
                    return r

            say("OPT-125m, the reference log's configuration")
            out["reference_log_opt125m_w3exp_a8"] = graph_or_eager(**BB.REFERENCE_LOG)
            # MEASURED at the reference's schedule (128 samples x 20 epochs, run.sh:6) under fp16 autocast, the W4A4
            # default (main.py:274); the fp32 path is measured at 128 samples x 2 epochs (its 20-epoch figure is a projection)
            say("Llama-2-7B block, fp16 autocast, 128 samples x 20 epochs")
            out["llama2_7b_w4a4_amp_128x20"] = graph_or_eager(model="7b", samples=128, epochs=20, amp=True, act="per_tensor")
            say("Llama-2-7B block, fp32, 128 samples x 2 epochs")
            out["llama2_7b_w4a4_fp32_128x2"] = BB.run("7b", 128, 2, False, "per_tensor", dev=dev)
            return out
    except Exception as e:  # the headline line must survive a failure of this auxiliary measurement
        out["error"] = f"{type(e).__name__}: {e}"
        return out
    # N ranks.  A rank that failed while the others entered a collective would leave them waiting, so every rank first runs
    # the whole loop on its own (no collective: this is also the warm-up), then the ranks agree (one MIN all-reduce of
    # a flag) whether all of them got through, and only then run the data-parallel loops.
    ok, err = 1, None
    try:
        for amp in (False, True):
            say(f"every rank on its own ({'fp16 autocast' if amp else 'fp32'})")
            BB.run("7b", 2, 1, amp, "per_tensor", dev=dev, dp=None, warmup=False)
    except Exception as e:
        ok, err = 0, f"{type(e).__name__}: {e}"
        print(f"[bench rank {rank}] local block-calibration run failed: {err}", file=sys.stderr, flush=True)
    flag = torch.tensor([ok], dtype=torch.int32, device="cpu" if rehearsal else dev)
    torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
    if int(flag.item()) == 0:
        out["error"] = err or "another rank failed in its local warm-up run; the data-parallel runs were skipped on all ranks"
        return out
    try:  # (an error every rank raises at the same point, e.g. a bug in this path, must not cost the headline line)
        dp = DPContext.from_env()
        say(f"data parallel over {world} ranks, fp32")
        out["llama2_7b_w4a4_fp32"] = BB.run("7b", 4, 2, False, "per_tensor", dev=dev, dp=dp, warmup=False)
        say(f"data parallel over {world} ranks, fp16 autocast")
        out["llama2_7b_w4a4_amp"] = BB.run("7b", 4, 2, True, "per_tensor", dev=dev, dp=dp, warmup=False)
    except Exception as e:
        import traceback
        out["error"] = f"{type(e).__name__}: {e}"
        print(f"[bench rank {rank}] data-parallel block calibration failed:\n{traceback.format_exc()}", file=sys.stderr, flush=True)
    return out


def launch_ranks(n: int, argv):
    """`python bench.py --gpus N ...` as the driver calls it: start N ranks of this script on this node through
    torch.distributed.run in a CHILD process (this process never initialises a GPU), pass rank 0's JSON line through
    and return the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in proc.stdout.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)  # anything else the ranks printed
    if lines:
        print(lines[-1])
    elif proc.returncode == 0:
        print("bench.py: the ranks finished without a result line", file=sys.stderr)
        return 1
    return proc.returncode


def main():
    if os.environ.get("MIQ_BENCH_STACKS"):  # every rank dumps its Python stacks every N seconds (where does a rank wait?)
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["MIQ_BENCH_STACKS"]), repeat=True, file=sys.stderr)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-block-calib", action="store_true")
    ap.add_argument("--block-calib", action="store_true",
                    help="one-GPU rehearsal of N > 1 ranks (MIQ_BENCH_ONE_GPU=1): also run the data-parallel block "
                         "calibration (it is on by default on real ranks, off in the rehearsal where N blocks share one GPU)")
    ap.add_argument("--no-families", action="store_true", help="skip the exponent / mixed-bit families object")
    ap.add_argument("--eager", action="store_true", help="launch kernel by kernel instead of replaying HIP graphs")
    ap.add_argument("--workload", choices=("c2", "c4"), default="c2",
                    help="c2 (default): the Llama-2-7B W4A4 calibration-step workload the metric is quoted on; c4: Llama-2-13B "
                         "mixed-bit weight quantization with the rows sharded over the ranks")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as a plain script: become the launcher.  Nothing has touched the GPU yet in this process.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU")
    # MIQ_BENCH_ONE_GPU=1 is a development aid: all ranks share GPU 0 and talk over gloo, which lets the
    # N > 1 code path be rehearsed on a single-GPU box.  The real launch is one rank per GPU over RCCL.
    rehearsal = os.environ.get("MIQ_BENCH_ONE_GPU") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            import datetime
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            torch.distributed.init_process_group("gloo", timeout=datetime.timedelta(seconds=60))
        else:
            torch.distributed.init_process_group("nccl", device_id=torch.device(dev))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if args.workload == "c4":
        run_c4(args, world, rank, dev, rehearsal, barrier)
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    wl = Workload(dev, seed=2 + 1000 * rank)
    fams = wl.families() if args.eager else wl.capture()
    idle = Segments()
    for _ in range(args.warmup):
        wl.step(fams, idle, False, world)
    # instrumented pass (outside the timed region): every family between two boundary events -> the family table, and
    # which family is the dominant one
    seg_all = Segments()
    for _ in range(args.steps):
        wl.step(fams, seg_all, True, world, boundaries=True)
    torch.cuda.synchronize()
    totals_all = seg_all.totals_ms()
    # the dominant KERNEL: the row families are one kernel each (two launches, one per row width), the activation
    # families chains of two kernels per call whose streaming kernels each take less than k_uniform_bwd_batch does
    # (profiles/: rocprofv3 kernel trace of this command)
    kernels_in = {"act_calib_quant_fwd": 2, "act_bwd": 2}
    dom = max(totals_all, key=lambda k: totals_all[k] / kernels_in.get(k, 1))
    # what the step loses when a family is left out: the sequence of all families against the sequence without it, each
    # between ONE pair of events over all steps (no event inside the sequence; the launch gaps of the family count)
    def sequence_ms(skip):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(args.steps + 2):
            if i == 2:
                a.record()
            for name, fn in fams:
                if name != skip:
                    fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / args.steps
    whole_ms = sequence_ms(None)
    marginal_ms = {name: whole_ms - sequence_ms(name) for name, _fn in fams}
    seg = Segments(only={dom})
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step(fams, seg, True, world)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    totals_local = seg.totals_ms()
    by_local = wl.bytes()
    launches = wl.launches()
    del wl, fams
    torch.cuda.empty_cache()
    fam34 = None
    if rank == 0 and not args.no_families:
        try:
            fam34 = families_c3_c4(dev)
        except Exception as e:  # the headline line must survive a failure of this auxiliary measurement
            fam34 = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    if world > 1:
        barrier()

    def emit(blk):
        """Rank 0's result line."""
        by = by_local
        step_bytes = sum(by.values())
        totals = totals_local  # the timed region: the dominant family only
        fam = {k: {"kernel": FAMILY_KERNEL[k], "GBps": round(by[k] * args.steps / (totals_all[k] * 1e-3) / 1e9, 1),
                   "frac_hbm_peak": round(by[k] * args.steps / (totals_all[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                   "ms_per_step": round(totals_all[k] / args.steps, 4), "launches_per_step": launches[k],
                   "ms_left_out": round(marginal_ms[k], 4),
                   "frac_hbm_peak_left_out": round(by[k] / (marginal_ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if marginal_ms[k] > 0 else None}
               for k in by}
        n_launch = launches[dom] * args.steps
        achieved = by[dom] * args.steps / (totals[dom] * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters (FETCH_SIZE doubled, WRITE_SIZE as read) of a separate rocprofv3 run of
        # this command by the builder; the file names that run
        traffic, traffic_source, traffic_current = None, None, None
        tpath = os.path.join(REPO, "profiles", "traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            per_step = tj.get(dom)
            traffic = per_step // launches[dom] if per_step else None
            traffic_source = tj.get("_source")
            # the counters were collected on one build of the kernels: say whether it is the one that just ran
            from miq import _lib
            traffic_current = tj.get("_csrc_digest") == _lib.source_digest()
        out = {
            "metric": "fake-quant GB/s (% HBM peak) + OmniQuant block-calib s/layer, Llama-2-7B",
            "value": round(step_bytes * world * args.steps / dt / 1e9, 1),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "Llama-2-7B block, W4A4, static per-cluster activation quant + RPTQ reorder gather, "
                                   "one 2048-token calibration sample per rank per step: quantizer kernels fwd+bwd "
                                   "(7 weights fp32, 10 activation sites fp32/fp16 in 5 calls, 2 gathers)",
                       "tokens_per_step_per_gpu": T, "w_bits": W_BITS, "a_bits": A_BITS,
                       "launch": "eager" if args.eager else "hipGraph replay per kernel family",
                       "parallelism": f"dp{world}" + (" + RCCL all-reduce of 84992 LWC grads" if world > 1 else "")},
            "frac_hbm_peak": round(step_bytes * args.steps / dt / 1e9 / HBM_PEAK_GBS, 4),
            "roofline": {"bound": "hbm", "kernel": FAMILY_KERNEL[dom], "family": dom, "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_source, "traffic_measured_on_this_build": traffic_current,
                         "bytes_per_launch": by[dom] // launches[dom],
                         "avg_launch_us": round(totals[dom] * 1e3 / n_launch, 2)},
            "families": fam,
            "families_timing": f"instrumented pass of {args.steps} steps in front of the timed region: the same graphs in the same "
                               "order with one HIP event at each of the 7 family boundaries (an event record takes about 5 us "
                               "of stream time, which every family figure includes once); the timed region itself brackets "
                               f"only {dom}, from which `roofline` is computed.  ms_left_out / frac_hbm_peak_left_out: the whole "
                               f"sequence ({whole_ms:.4f} ms between one pair of events, no event inside) minus the sequence "
                               "without that family, i.e. what the family costs the step with its launch gaps and without "
                               "any event record",
        }
        if fam34 is not None:
            out["families_c3_c4"] = fam34
        if blk is not None:
            out["block_calib"] = blk
        if world > 1:
            out["config"]["note"] = ("no hardware scaling curve exists in this repository: the multi-GPU numbers are whatever "
                                     "this run measures" + ("; THIS run is a one-GPU rehearsal over gloo (all ranks share GPU 0): "
                                                            "functional check only" if rehearsal else ""))
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(min(os.cpu_count() or 1, 16))
        print(json.dumps(out))

    want_blk = not args.no_block_calib and (world == 1 or not rehearsal or args.block_calib)
    blk = None
    if want_blk and world > 1:
        # The data-parallel block calibration runs collectives in a loop.  Should a rank ever wait in one for good, the
        # result line must still appear: past the deadline rank 0 prints it with an error object in place of the
        # measurement, and every rank leaves without waiting for the others.
        import threading
        finished = threading.Event()
        deadline = float(os.environ.get("MIQ_BENCH_BLOCK_DEADLINE", "420"))

        def watchdog():
            if finished.wait(deadline):
                return
            if rank == 0:
                emit({"error": f"the data-parallel block calibration did not finish within {deadline:.0f} s; the run was ended"})
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)

        threading.Thread(target=watchdog, daemon=True).start()
        blk = block_calib(dev, world, rank, rehearsal)
        finished.set()
    elif want_blk:
        blk = block_calib(dev, world, rank, rehearsal)
    if rank == 0:
        emit(blk)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
