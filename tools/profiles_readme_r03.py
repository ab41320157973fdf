#!/usr/bin/env python3
"""Writes the round-3 section of profiles/README.md from the installed artifacts of a measurement tag, above the
round-2 section (everything from '## Round 2' on is kept as it is):  python3 tools/profiles_readme_r03.py r03_a"""
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
    b2 = last_json(prof + "r02_g_bench.json.log")
    ks = [(r["Name"].split("(")[0].replace("void miq::", "").replace("miq::", ""), int(r["Calls"]), float(r["AverageNs"]) / 1e3)
          for r in csv.DictReader(open(prof + f"{tag}_bench_kernel_stats.csv")) if "miq::" in r["Name"]]
    traffic = json.load(open(prof + "traffic.json"))
    L = ["# profiles/\n"]
    L.append(f"MI355X, gfx950, ROCm 7.2, one GPU.  `{tag}_*` is round 3's final build; `r02_*` and `r01_*` are the earlier rounds' files, "
             "described further down.  `tools/collect_on_gpu.sh <dir>` runs every measurement on the GPU box, "
             "`tools/collect_profiles.py <dir> <tag>` installs it here, `tools/profiles_readme_r03.py <tag>` writes this section.\n")
    L.append("## Round 3\n")
    L.append("Same files as round 2 under the new tag (table in the next section: bench line, kernel trace, the two PMC passes and "
             "`traffic.json`, SQ counters, families, row-sharded workload, blocks, micro-benchmarks), plus `"
             f"{tag}_bench_let.txt` (`tools/bench_let.py`) and `{tag}_softmax_time.txt`.  `traffic.json` now carries `_csrc_digest`, the "
             "digest of the kernel sources it was measured on; `bench.py` reports `traffic_measured_on_this_build` from it.  The PMC "
             "passes run `bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-block-calib --no-families`.\n")
    L.append(f"`value` = {b['value']} GB/s algorithmic over the whole step ({b['ms_per_step']} ms, {b['frac_hbm_peak'] * 100:.1f} % of the 8 TB/s "
             f"peak); round 2: {b2['value']} GB/s ({b2['ms_per_step']} ms).\n")
    L.append("Family table: `ms / step` and `of 8 TB/s` are taken between two HIP events of the family's own in an instrumented pass in "
             "front of the timed region (an event record costs about 5 µs of stream time; the timed region brackets only the dominant "
             "family); `left out` is what the family costs the step: the whole sequence between one pair of events minus the "
             "sequence without the family.\n")
    L.append("| family | ms / step | GB/s | of 8 TB/s | left out: ms, of 8 TB/s | launches | PMC bytes / algorithmic bytes per step | round 2: of 8 TB/s, launches |\n|---|---|---|---|---|---|---|---|")
    for k, v in b["families"].items():
        o = b2["families"][k]
        alg = v["GBps"] * 1e9 * v["ms_per_step"] * 1e-3
        lo = v.get("frac_hbm_peak_left_out")
        left = f"{v['ms_left_out']}, {lo * 100:.1f} %" if lo else "-"
        L.append(f"| `{k}` ({v['kernel']}) | {v['ms_per_step']} | {v['GBps']} | {v['frac_hbm_peak'] * 100:.1f} % | {left} | {v['launches_per_step']} | "
                 f"{traffic[k] / 1e6:.0f} MB / {alg / 1e6:.0f} MB | {o['frac_hbm_peak'] * 100:.1f} %, {o['launches_per_step']} |")
    rf = b["roofline"]
    L.append(f"\n`roofline`: dominant family `{rf['family']}` (kernel `{rf['kernel']}`), {rf['achieved']} GB/s = {rf['frac']} of peak, "
             f"{rf['avg_launch_us']} µs per launch by HIP events; `cpu_baseline`: {b['cpu_baseline']['value']} GB/s on "
             f"{b['cpu_baseline']['cores']} host cores (oracle, `kind: port`).  `roofline.traffic` = {rf['traffic']} B per launch from the "
             f"PMC passes of this tag (`traffic_measured_on_this_build: {str(rf['traffic_measured_on_this_build']).lower()}`) against "
             f"{rf['bytes_per_launch']} B algorithmic.\n")
    L.append("What changed the families: outputs of 16 MiB and more are written non-temporally (their write-back no longer falls on the "
             "next kernels: `act_calib_quant_fwd` 0.49 → 0.58; the row backward pays 1–2 % for it), the activation backward has no partial "
             "planes any more (`k_colbwd_stream` + `k_colbwd_pairs`), the two gathers of a step share one launch and stream their output; "
             "the timed region lost ten of its twelve event records per step (1.14 → 1.07 ms); the exponent v1 grid is batched per row "
             "width like the others, and the 512-lane backward tiles of the exponent and mixed-bit grids are capped at 128 registers "
             "(two blocks per CU instead of one).\n")
    L.append("Quantizer families on the C3 / C4 shapes (`families_c3_c4` of the bench line; 20 back-to-back graph replays between one pair of events, median of three rounds):\n")
    L.append("| family | kernel | shapes | GB/s | of 8 TB/s | launches | µs per launch |\n|---|---|---|---|---|---|---|")
    for k, v in b["families_c3_c4"].items():
        L.append(f"| `{k}` | {v['kernel']} | {v['shapes']} | {v['GBps']} | {v['frac_hbm_peak'] * 100:.1f} % | {v['launches']} | {v['us_per_launch']} |")
    L.append("\nKernel trace of the bench (`" + tag + "_bench_kernel_stats.csv`; µs per launch incl. the gap to the previous kernel in graph replay):\n")
    L.append("| kernel | calls | µs |\n|---|---|---|")
    for name, calls, us in ks[:16]:
        L.append(f"| `{name}` | {calls} | {us:.1f} |")
    L.append("\n### Block calibration (`block_calib` of the bench line)\n")
    L.append("| run | step | graph steps / steps | s per step | s per layer (this run) | notes |\n|---|---|---|---|---|---|")
    for k, v in b["block_calib"].items():
        if isinstance(v, dict) and "s_per_step" in v:
            extra = f"; reference log {v['reference_s_per_layer']} s/layer → {v['speedup_vs_reference_log']}×" if "reference_s_per_layer" in v else ""
            L.append(f"| `{k}` | {v['step']} | {v['graph_steps']} / {v['steps']} | {v['s_per_step']} | {v['s_layer_this_run']}{extra} | {v['notes'] or '—'} |")
    L.append("")
    path = prof + "README.md"
    old = open(path).read()
    rest = old[old.index("## Round 2"):]
    open(path, "w").write("\n".join(L) + "\n" + rest)


if __name__ == "__main__":
    main()
