#!/usr/bin/env python3
"""GB/s of algorithmic bytes for each operator family on the configurations the baseline names:

  C1  OPT-125m W3A16 + LWC        uniform row quantizer on the six OPT-125m weight shapes, fwd and fwd+bwd
  C2  Llama-2-7B W4A4 per-cluster  static activation calibrate+quantize / eval quantize / backward, reorder gather
  C3  Llama-2-7B W3A16 exponent    exp-v2 and exp-v1 row quantizers on the seven 7B weight shapes, fwd and bwd
  C4  Llama-2-13B mixed bit        learnable mixed-bit (soft and hard) and fixed mixed-bit on the seven 13B shapes

Every kernel sequence of a configuration is captured in one HIP graph over rotating buffer sets that together
exceed the 256 MiB Infinity Cache, and the graph is timed with HIP events (10 back-to-back replays between one pair
of events, median of three rounds).  Algorithmic
bytes: forward 2ns, backward 3ns (2ns without dx), calibrate+quantize 3ns, gather 2ns.
Prints one JSON object; --out writes it to a file.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "llm-quantization_amd"))
from miq import functional as F  # noqa: E402

DEV = "cuda:0"
PEAK = 8000.0
OPT125 = [(768, 768)] * 4 + [(3072, 768), (768, 3072)]
L7 = [(4096, 4096)] * 4 + [(11008, 4096)] * 2 + [(4096, 11008)]
L13 = [(5120, 5120)] * 4 + [(13824, 5120)] * 2 + [(5120, 13824)]


def time_graph(fn, reps=10, rounds=3):
    # warm up ON the capture stream: per-stream state of the library (the activation kernels' statistics table) must
    # exist before the capture, which refuses to allocate it
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        fn()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):  # an event record costs about 5 us of stream time: `reps` replays share one pair (as in bench.py)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3 / reps)
    return sorted(ts)[len(ts) // 2]


def weights(shapes, dt, copies):
    sets = []
    for c in range(copies):
        sets.append([(torch.randn(o, i, device=DEV, dtype=dt) * 0.02, torch.randn(o, i, device=DEV, dtype=dt) * 1e-3,
                      torch.full((o, 1), 0.982, device=DEV, dtype=dt), torch.full((o, 1), 0.982, device=DEV, dtype=dt))
                     for (o, i) in shapes])
    return sets


def copies_for(shapes, esize, arrays=2):
    one = sum(o * i for o, i in shapes) * esize * arrays
    return max(1, int(600e6 // one) + 1)


def row_family(name, shapes, dt, fwd, bwd, results, bwd_nodx=None):
    es = 4 if dt == torch.float32 else 2
    sets = weights(shapes, dt, copies_for(shapes, es))
    n = sum(o * i for o, i in shapes) * es * len(sets)
    for tag, fn, mult in (("fwd", fwd, 2), ("bwd", bwd, 3), ("bwd_no_dx", bwd_nodx, 2)):
        if fn is None:
            continue
        t = time_graph(lambda: [fn(*w) for s in sets for w in s])
        gbs = mult * n / t / 1e9
        results[f"{name}.{tag}.{'f32' if es == 4 else 'f16'}"] = {"GBps": round(gbs, 1), "frac_hbm_peak": round(gbs / PEAK, 4),
                                                                "us_per_matrix_set": round(t / len(sets) * 1e6, 1)}
    del sets
    torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    res = {}
    f32, f16 = torch.float32, torch.float16
    # C1
    row_family("C1.uniform_w3_lwc.opt125m", OPT125, f32,
               lambda x, g, u, l: F.rowquant_uniform_fwd(x, 3, False, u, l),
               lambda x, g, u, l: F.rowquant_uniform_bwd(g, x, 3, False, u, l, want_dx=True), res,
               lambda x, g, u, l: F.rowquant_uniform_bwd(g, x, 3, False, u, l, want_dx=False))
    # the same six matrices through the batched entry points: one launch per row width instead of one per matrix
    def c1_batched():
        es = 4
        sets = weights(OPT125, f32, copies_for(OPT125, es))
        n = sum(o * i for o, i in OPT125) * es * len(sets)
        groups = {}
        for j, (_o, i) in enumerate(OPT125):
            groups.setdefault(i, []).append(j)

        def fwd():
            for st in sets:
                for ids in groups.values():
                    F.rowquant_uniform_fwd_batch([st[j][0] for j in ids], 3, False, [st[j][2] for j in ids], [st[j][3] for j in ids])

        def bwd():
            for st in sets:
                for ids in groups.values():
                    F.rowquant_uniform_bwd_batch([st[j][1] for j in ids], [st[j][0] for j in ids], 3, False, [st[j][2] for j in ids],
                                                 [st[j][3] for j in ids])

        for tag, fn, mult in (("fwd", fwd, 2), ("bwd", bwd, 3)):
            t = time_graph(fn)
            gbs = mult * n / t / 1e9
            res[f"C1.uniform_w3_lwc.opt125m.batched.{tag}.f32"] = {"GBps": round(gbs, 1), "frac_hbm_peak": round(gbs / PEAK, 4),
                                                                  "us_per_matrix_set": round(t / len(sets) * 1e6, 1)}
        del sets
        torch.cuda.empty_cache()

    c1_batched()
    # uniform on 7B shapes (the weights of C2)
    for dt in (f32, f16):
        row_family("C2.uniform_w4_lwc.llama7b", L7, dt,
                   lambda x, g, u, l: F.rowquant_uniform_fwd(x, 4, False, u, l),
                   lambda x, g, u, l: F.rowquant_uniform_bwd(g, x, 4, False, u, l, want_dx=True), res,
                   lambda x, g, u, l: F.rowquant_uniform_bwd(g, x, 4, False, u, l, want_dx=False))
    # C3
    for dt in (f32, f16):
        row_family("C3.exp2_w3_lwc.llama7b", L7, dt,
                   lambda x, g, u, l: F.rowquant_exp2_fwd(x, 3, u, l),
                   lambda x, g, u, l: F.rowquant_exp2_bwd(g, x, 3, u, l, want_dx=True), res,
                   lambda x, g, u, l: F.rowquant_exp2_bwd(g, x, 3, u, l, want_dx=False))
        row_family("C3.exp1_w3_lwc.llama7b", L7, dt,
                   lambda x, g, u, l: F.rowquant_exp1_fwd(x, 3, True, u, l),
                   lambda x, g, u, l: F.rowquant_exp1_bwd(g, x, 3, True, u, l, want_dx=True), res)
    # C3 through the batched entry points: the seven Llama-2-7B matrices in one launch per row width
    def c3_batched():
        es = 4
        sets = weights(L7, f32, copies_for(L7, es))
        n = sum(o * i for o, i in L7) * es * len(sets)
        groups = {}
        for j, (_o, i) in enumerate(L7):
            groups.setdefault(i, []).append(j)

        def fwd():
            for st in sets:
                for ids in groups.values():
                    F.rowquant_exp2_fwd_batch([st[j][0] for j in ids], 3, [st[j][2] for j in ids], [st[j][3] for j in ids])

        def bwd():
            for st in sets:
                for ids in groups.values():
                    F.rowquant_exp2_bwd_batch([st[j][1] for j in ids], [st[j][0] for j in ids], 3, [st[j][2] for j in ids],
                                              [st[j][3] for j in ids])

        for tag, fn, mult in (("fwd", fwd, 2), ("bwd", bwd, 3)):
            t = time_graph(fn)
            gbs = mult * n / t / 1e9
            res[f"C3.exp2_w3_lwc.llama7b.batched.{tag}.f32"] = {"GBps": round(gbs, 1), "frac_hbm_peak": round(gbs / PEAK, 4),
                                                               "us_per_matrix_set": round(t / len(sets) * 1e6, 1)}
        del sets
        torch.cuda.empty_cache()

    c3_batched()
    # C4
    bits = {}

    def nb(x, hard):
        key = (x.shape[0], hard)
        if key not in bits:
            b = 3 + 0.5 + torch.rand(x.shape[0], 1, device=DEV) / 4
            bits[key] = b.round() if hard else b
        return bits[key]

    for hard in (False, True):
        row_family(f"C4.mix_{'hard' if hard else 'soft'}_lwc.llama13b", L13, f32,
                   lambda x, g, u, l: F.rowquant_mix_fwd(x, nb(x, hard), False, u, l, hard=hard),
                   lambda x, g, u, l: F.rowquant_mix_bwd(g, x, nb(x, hard), False, u, l, hard=hard, want_dx=True), res)
    # C4 through the batched entry points: the seven Llama-2-13B matrices in one launch per row width
    def c4_batched(hard):
        es = 4
        sets = weights(L13, f32, copies_for(L13, es))
        n = sum(o * i for o, i in L13) * es * len(sets)
        groups = {}
        for j, (_o, i) in enumerate(L13):
            groups.setdefault(i, []).append(j)

        def fwd():
            for st in sets:
                for ids in groups.values():
                    F.rowquant_mix_fwd_batch([st[j][0] for j in ids], [nb(st[j][0], hard) for j in ids], False,
                                             [st[j][2] for j in ids], [st[j][3] for j in ids], hard=hard)

        def bwd():
            for st in sets:
                for ids in groups.values():
                    F.rowquant_mix_bwd_batch([st[j][1] for j in ids], [st[j][0] for j in ids], [nb(st[j][0], hard) for j in ids],
                                             False, [st[j][2] for j in ids], [st[j][3] for j in ids], hard=hard)

        for tag, fn, mult in (("fwd", fwd, 2), ("bwd", bwd, 3)):
            t = time_graph(fn)
            gbs = mult * n / t / 1e9
            res[f"C4.mix_{'hard' if hard else 'soft'}_lwc.llama13b.batched.{tag}.f32"] = {
                "GBps": round(gbs, 1), "frac_hbm_peak": round(gbs / PEAK, 4), "us_per_matrix_set": round(t / len(sets) * 1e6, 1)}
        del sets
        torch.cuda.empty_cache()

    for hard in (False, True):
        c4_batched(hard)
    fixed = {}

    def fb(x):
        if x.shape[0] not in fixed:
            r = torch.rand(x.shape[0], device=DEV)
            fixed[x.shape[0]] = torch.where(r < 0.1, 4.0, torch.where(r < 0.9, 3.0, 2.0)).float()
        return fixed[x.shape[0]]

    for dt in (f32, f16):
        row_family("C4.mix2_fixed_lwc.llama13b", L13, dt,
                   lambda x, g, u, l: F.rowquant_uniform_fwd(x, 4, False, u, l, 0, fb(x), True),
                   lambda x, g, u, l: F.rowquant_uniform_bwd(g, x, 4, False, u, l, 0, fb(x), True, want_dx=True), res)
    # C2 activations
    T = 2048
    for dt, es in ((f32, 4), (f16, 2)):
        for C, counts in ((4096, [128] * 32), (11008, [344] * 32)):
            copies = max(3, (int(600e6 // (T * C * es * 2)) + 3) // 3 * 3)
            xs = [torch.randn(T, C, device=DEV, dtype=dt) for _ in range(copies)]
            gs = [torch.randn(T, C, device=DEV, dtype=dt) * 1e-3 for _ in range(copies)]
            lay = F.ClusterLayout.get(counts, C, torch.device(DEV))
            n = T * C * es * copies
            saved = []

            def fwd():  # one slab per call: the shape of a site that stands alone in a block
                saved.clear()
                for x in xs:
                    saved.append(F.act_fresh_quant_fwd([x], 4, lay)[2])

            def bwd():
                for x, g, vecs in zip(xs, gs, saved):
                    F.act_fresh_quant_bwd([g], [x], vecs, 4, lay)

            saved3 = []

            def fwd3():  # three slabs per call: the q / k / v sites of a block
                saved3.clear()
                for i in range(0, len(xs) - 2, 3):
                    saved3.append(F.act_fresh_quant_fwd(xs[i:i + 3], 4, lay)[2])

            def bwd3():
                for j, i in enumerate(range(0, len(xs) - 2, 3)):
                    F.act_fresh_quant_bwd(gs[i:i + 3], xs[i:i + 3], saved3[j], 4, lay)

            def evalq():
                for x, vecs in zip(xs, saved):
                    F.colquant_fwd(x, vecs[0][2], vecs[0][4], 4)

            idx = torch.randperm(C, device=DEV).to(torch.int32)

            def gather():
                for x in xs:
                    F.gather_lastdim(x, idx)

            tag = f"{'f32' if es == 4 else 'f16'}.{T}x{C}"
            n3 = T * C * es * (copies // 3 * 3)
            for name, fn, mult in (("C2.act_calibrate_quantize", fwd, 3), ("C2.act_backward", bwd, 3),
                                   ("C2.act_calibrate_quantize_x3", fwd3, 3 * n3 / n), ("C2.act_backward_x3", bwd3, 3 * n3 / n),
                                   ("C2.act_eval_quantize", evalq, 2), ("C2.reorder_gather", gather, 2)):
                t = time_graph(fn)
                gbs = mult * n / t / 1e9
                res[f"{name}.{tag}"] = {"GBps": round(gbs, 1), "frac_hbm_peak": round(gbs / PEAK, 4), "us_per_tensor": round(t / copies * 1e6, 1)}
            del xs, gs
            torch.cuda.empty_cache()
    # reference point: a plain device copy of the same volume
    x = torch.randn(64 * 1024 * 1024, device=DEV)
    y = torch.empty_like(x)
    t = time_graph(lambda: y.copy_(x))
    res["reference.torch_copy_256MiB"] = {"GBps": round(2 * x.numel() * 4 / t / 1e9, 1), "frac_hbm_peak": round(2 * x.numel() * 4 / t / 1e9 / PEAK, 4)}
    print(json.dumps(res, indent=1))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
