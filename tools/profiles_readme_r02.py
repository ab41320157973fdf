#!/usr/bin/env python3
"""Rewrites the round-2 half of profiles/README.md from the installed artifacts of a measurement tag (the round-1 half,
below the horizontal rule, is kept as it is):  python3 tools/profiles_readme_r02.py r02_c"""
import csv
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_json(path):
    return json.loads([l for l in open(path) if l.startswith("{")][-1])


def main():
    tag = sys.argv[1]
    prof = os.path.join(REPO, "profiles") + "/"
    b = last_json(prof + f"{tag}_bench.json.log")
    b1 = last_json(prof + "r01_h_bench.json.log")
    ks = {r["Name"].split("(")[0].replace("void miq::", "").replace("miq::", ""): (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
          for r in csv.DictReader(open(prof + f"{tag}_bench_kernel_stats.csv")) if "miq::" in r["Name"]}
    pk = json.load(open(prof + f"{tag}_pmc_traffic_per_kernel.json"))["kernels"]
    sq = json.load(open(prof + f"{tag}_sq_bench.json"))
    sqr = json.load(open(prof + f"{tag}_sq_rows.json"))
    c4 = last_json(prof + f"{tag}_bench_c4.json.log")
    fam = json.load(open(prof + f"{tag}_families.json"))
    fam1 = json.load(open(prof + "r01_h_families.json"))
    L = ["# profiles/\n"]
    L.append(f"MI355X, gfx950, ROCm 7.2, one GPU.  `{tag}_*` is round 2's final build (`r02_a_*`, `r02_b_*`, `r02_c_*`, `r02_f_*`: earlier states of the "
             "round, kept for the history: before the statistics-kernel / cluster-kernel / division work, before the graph-replayed "
             "calibration step, before the launch options left the element loops of the fp16 row, exponent, LET and "
             "attention-score kernels, and before the exponent and mixed-bit quantizers took several matrices per launch); `r01_*` are round 1's files, described in the second half of this file.  "
             "`tools/collect_on_gpu.sh <dir>` runs every measurement on the GPU box, `tools/collect_profiles.py <dir> <tag>` installs it "
             "here, `tools/profiles_readme_r02.py <tag>` writes this half of the file.\n")
    L.append("## Round 2\n")
    L.append("| file | what |\n|---|---|")
    L.append(f"| `{tag}_bench.json.log` | `python3 bench.py` (default flags): the result line incl. `roofline`, `cpu_baseline`, `block_calib` (Llama-2-7B block measured at 128 samples x 20 epochs under fp16 autocast) |")
    L.append(f"| `{tag}_bench_kernel_stats.csv` | `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-block-calib` |")
    L.append(f"| `{tag}_pmc_fetch_size.csv`, `{tag}_pmc_write_size.csv`, `{tag}_pmc_traffic_per_kernel.json`, `traffic.json` | `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, no tracing) on `python3 bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-block-calib`; `tools/pmc_summary.py`: HBM bytes per launch = 2·FETCH_SIZE·1024 + WRITE_SIZE·1024; `traffic.json` is what `bench.py` reads for `roofline.traffic` |")
    L.append(f"| `{tag}_sq_rows.json`, `{tag}_sq_bench.json` | `rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES` on `tools/row_pmc.py` (uniform / exponent v1, v2 / mixed-bit row quantizers, fp32 and fp16, forward and backward) and on the eager bench; `tools/row_pmc_summary.py`: per kernel the share of wave cycles issuing any / VALU instructions, parked (`s_waitcnt`, barriers), stalled at issue, VALU instructions per wave |")
    L.append(f"| `{tag}_families.json` | `tools/bench_families.py`: every operator family on the baseline's named configurations (`_x3` = three slabs per call, `...batched` = the matrices of a layer through the batched entry points, one launch per row width) |")
    L.append(f"| `{tag}_bench_c4.json.log` | `python3 bench.py --workload c4` on one GPU: the row-sharded Llama-2-13B mixed-bit workload (BASELINE configs[3]) |")
    L.append(f"| `{tag}_blocks_extra.jsonl`, `{tag}_block7b_kernel_stats.csv`, `{tag}_ab_block.json` | `tools/bench_block.py` for the 13B / 70B shaped blocks, kernel trace of the Llama-2-7B fp16 calibration step, `tools/ab_block.py` |")
    L.append(f"| `{tag}_norm_quant_time.txt` | `tools/dev/norm_quant_time.py`: norm -> activation quantizer as two functions against the norm kernels that publish the statistics themselves |")
    L.append(f"| `{tag}_ubench_actbench.txt`, `{tag}_ubench_readbw.txt`, `{tag}_ubench_statbench.txt` | `tools/ubench/*.hip`: the activation forward per slab and slabs per launch against the four-kernel form; pure read / copy ceilings at the activation sizes; statistics-kernel variants |\n")
    L.append(f"`value` = {b['value']} GB/s algorithmic over the whole step ({b['ms_per_step']} ms, {b['frac_hbm_peak'] * 100:.1f} % of the 8 TB/s peak); round 1: {b1['value']} GB/s ({b1['ms_per_step']} ms).\n")
    L.append("| family | ms / step | GB/s | of 8 TB/s | launches | round 1: of 8 TB/s, launches |\n|---|---|---|---|---|---|")
    r1l = {'rowquant_fwd': 2, 'rowquant_bwd': 2, 'act_calib_quant_fwd': 40, 'act_bwd': 40, 'gather_fwd': 2, 'gather_bwd': 2}
    for k, v in b['families'].items():
        o = b1['families'][k]
        L.append(f"| `{k}` ({v['kernel']}) | {v['ms_per_step']} | {v['GBps']} | {v['frac_hbm_peak'] * 100:.1f} % | {v['launches_per_step']} | {o['frac_hbm_peak'] * 100:.1f} %, {r1l[k]} |")
    rf = b['roofline']
    ka = ks['k_uniform_bwd_batch<float, 4, 256, 4, true>'][1]
    kb = next(v[1] for k, v in ks.items() if k.startswith('k_uniform_bwd_batch<float, 4, 512,'))  # the [4096, 11008] tile
    L.append(f"\n`roofline`: dominant family `{rf['family']}` (kernel `{rf['kernel']}`), {rf['achieved']} GB/s = {rf['frac']} of peak, {rf['avg_launch_us']} µs per launch by HIP events "
             f"(kernel trace: {ka:.1f} µs for the six [*, 4096] matrices + {kb:.1f} µs for [4096, 11008] = {(ka + kb) / 2:.1f} µs average; in graph replay the trace's durations abut, i.e. include the launch gap), "
             f"PMC traffic {rf['traffic']} B vs {rf['bytes_per_launch']} B algorithmic per launch.  `cpu_baseline`: {b['cpu_baseline']['value']} GB/s on 16 host cores (oracle, `kind: port`).\n")
    L.append("Activation kernels in the bench trace (µs per launch incl. the gap to the previous kernel; calls in 27 traced steps):\n")
    L.append("| kernel | calls | µs |\n|---|---|---|")
    for k, (c, u) in ks.items():
        if 'col' in k:
            L.append(f"| `{k}` | {c} | {u:.1f} |")
    L.append("\nPMC bytes per launch of this library's kernels (`" + tag + "_pmc_traffic_per_kernel.json`; the largest activation call is three fp32 [2048, 4096] slabs = 100.7 MB: the statistics kernel reads them once, the quantize kernel reads them again and writes 100.7 MB):\n")
    L.append("| kernel | launches | HBM read MB | HBM write MB |\n|---|---|---|---|")
    for k in pk:
        L.append(f"| `{k['kernel'][:60]}` | {k['launches']} | {k['hbm_read_bytes_per_launch'] / 1e6:.1f} | {k['hbm_write_bytes_per_launch'] / 1e6:.1f} |")
    for title, d in ((f"SQ counters, bench kernels (`{tag}_sq_bench.json`; shares of wave cycles)", sq),
                     (f"SQ counters, row quantizers on Llama shapes (`{tag}_sq_rows.json`)", sqr)):
        L.append(f"\n{title}:\n")
        L.append("| kernel | issuing any | issuing VALU | parked | stalled at issue | VALU insts / wave | waves / SIMD (est.) | SIMD VALU busy (est.) |\n|---|---|---|---|---|---|---|---|")
        for k, v in d.items():
            L.append(f"| `{k}` | {v['active_any_frac']:.2f} | {v['active_valu_frac']:.2f} | {v['wait_any_frac']:.2f} | {v['wait_inst_any_frac']:.2f} | {v['valu_insts_per_wave']:.0f} "
                     f"| {v.get('waves_per_simd_est', float('nan')):.1f} | {v.get('simd_valu_busy_est', float('nan')):.2f} |")
    L.append("\nReading.  The first four columns are shares of WAVE cycles: a wave of these kernels issues VALU instructions for 5-30 % of its life and is parked on memory for 40-65 %.  "
             "The last two relate the same counters to the machine (dispatch duration x 1024 SIMDs x 2.4 GHz, `tools/row_pmc_summary.py`): with three to six waves resident per SIMD, the vector units of "
             "the exponent and mixed-bit kernels and of every fp16 row kernel are busy 70-98 % of the launch, those of the fp32 uniform kernels 50-80 %, those of the activation kernels 25-60 % at under three waves per SIMD.  "
             "So the exponent, mixed-bit and fp16 row kernels are bound by the vector unit (fewer instructions per element pay there: the per-row choice of a plain element function took the exponent v2 "
             "forward from 0.65 to 0.70 of peak and its backward from 0.556 to 0.62, the same with lane-level tie handling the mixed-bit backward from 0.517 to 0.553), the fp32 uniform kernels sit at the copy ceiling with the vector "
             "unit in second place (646 -> 511 VALU instructions per wave moved the forward by 2 %), and the activation kernels are bound by launch ramp and latency at their occupancy "
             "(running extremes instead of per-element tie counters: 0.3 %).\n")
    bc = b['block_calib']
    L.append("### Block calibration (`block_calib` of the bench line)\n")
    L.append("| configuration | step | s / optimizer step | s / layer, measured | note |\n|---|---|---|---|---|")
    o = bc['reference_log_opt125m_w3exp_a8']
    L.append(f"| reference's committed run log: OPT-125m, exp W3 sym, A8 per_cluster, LET+LWC, 2 epochs x 128 samples x 2048 tokens, fp32 | {o.get('step', 'eager')} | {o['s_per_step']} | **{o['s_layer_this_run']}** | reference: 12.86 s/layer on an unnamed NVIDIA GPU -> {o['speedup_vs_reference_log']}x |")
    o = bc['llama2_7b_w4a4_amp_128x20']
    L.append(f"| Llama-2-7B block, W4A4 per-tensor static, LET+LWC, fp16 autocast, 128 samples x 20 epochs (the reference schedule) | {o.get('step', 'eager')} | {o['s_per_step']} | **{o['s_layer_this_run']}** | loss {o['loss_first']:.3f} -> {o['loss_last']:.3f}; peak memory {o['peak_mem_GiB']} GiB |")
    o = bc['llama2_7b_w4a4_fp32_128x2']
    L.append(f"| same, fp32 (`--deactive_amp`), 128 samples x 2 epochs | {o.get('step', 'eager')} | {o['s_per_step']} | {o['s_layer_this_run']} (2 epochs) | loss {o['loss_first']:.3f} -> {o['loss_last']:.3f}; 20 epochs projected: {o['s_per_layer_projected_128x20']} s |")
    for l in open(prof + f"{tag}_blocks_extra.jsonl"):
        d = json.loads(l)
        L.append(f"| {d['model']}, W{d['w_bits']}A{d['a_bits']}, {d['precision']} | {d.get('step', 'eager')} | {d['s_per_step']} | — | projected 128 x 20: {d['s_per_layer_projected_128x20']} s/layer on one GPU; peak memory {d['peak_mem_GiB']} GiB |")
    L.append("\nThe graph-replayed step follows the eager trajectory bit for bit (checked while it runs; `tests/test_gpu_graph_step.py`); eager numbers of the same build: OPT-125m 3.98 ms per step, 1.23 s/layer; Llama-2-7B fp16 5.85 ms per step.\n")
    L.append(f"### Row-sharded Llama-2-13B mixed-bit workload, one GPU (`{tag}_bench_c4.json.log`)\n")
    L.append(f"`value` = {c4['value']} GB/s for the whole layer ({c4['ms_per_step']} ms per step: soft + hard learnable mixed-bit forward + backward and fixed mixed-bit forward + backward on the seven matrices).\n")
    L.append("| family | kernel | GB/s | of 8 TB/s |\n|---|---|---|---|")
    for k, v in c4['families'].items():
        L.append(f"| `{k}` | `{v['kernel']}` | {v['GBps_per_rank']} | {v['frac_hbm_peak'] * 100:.1f} % |")
    L.append("\nNo multi-GPU curve exists for it yet (the driver's scaling run uses the default workload); the N > 1 path is covered by the gloo tests and the one-GPU rehearsal.\n")
    L.append(f"### Operator families (`{tag}_families.json`; round 1 in the last column)\n")
    L.append("| configuration . pass . dtype | GB/s | of 8 TB/s | round 1 |\n|---|---|---|---|")
    for k, v in fam.items():
        o = fam1.get(k, {}).get('frac_hbm_peak')
        L.append(f"| `{k}` | {v['GBps']} | {v['frac_hbm_peak'] * 100:.1f} % | {'' if o is None else f'{o * 100:.1f} %'} |")
    L.append(f"\n### Norm followed by the activation quantizer (`{tag}_norm_quant_time.txt`)\n")
    txt = "\n".join(l for l in open(prof + f"{tag}_norm_quant_time.txt").read().splitlines() if l.startswith(("rms", "ln")))
    L.append("```\n" + txt + "\n```\n")
    L.append("The norm kernels that publish the statistics of their own output save the statistics kernel (≈8 µs: it reads the norm output from cache) and pay for it with 64 atomics and 72 cross-lane exchanges per row when "
             "there are 32 clusters; only per-tensor statistics (one slot, one reduction per row) break even, and only they take this path by default.\n")
    old = open(prof + "README.md").read()
    tail = old[old.index("\n---\n\n## Round 1"):]
    open(prof + "README.md", "w").write("\n".join(L) + tail)


if __name__ == "__main__":
    main()
