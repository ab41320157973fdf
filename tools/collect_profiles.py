#!/usr/bin/env python3
"""Installs one round's measurement directory (as written by the collection command in profiles/README.md)
into profiles/: bench line, kernel-trace summary, the library's rows of the two PMC passes, the per-kernel
traffic summary, the per-family traffic file that bench.py reads, and the family table.

usage: collect_profiles.py <gpurun_out/dir> <tag, e.g. r01_h>
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src, tag = sys.argv[1], sys.argv[2]
    prof = os.path.join(REPO, "profiles")
    shutil.copy(os.path.join(src, "bench.json.log"), os.path.join(prof, f"{tag}_bench.json.log"))
    shutil.copy(glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))[0], os.path.join(prof, f"{tag}_bench_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "families.json"), os.path.join(prof, f"{tag}_families.json"))
    extra = os.path.join(src, "blocks_extra.jsonl")
    if os.path.exists(extra):
        shutil.copy(extra, os.path.join(prof, f"{tag}_blocks_extra.jsonl"))
    for name, dst in (("sq_rows.json", "sq_rows.json"), ("sq_bench.json", "sq_bench.json"), ("bench_c4.json.log", "bench_c4.json.log"),
                      ("norm_quant_time.txt", "norm_quant_time.txt"), ("ubench_actbench.txt", "ubench_actbench.txt"),
                      ("ubench_readbw.txt", "ubench_readbw.txt"), ("ubench_statbench.txt", "ubench_statbench.txt"),
                      ("ab_block.json", "ab_block.json"), ("bench_let.txt", "bench_let.txt"),
                      ("softmax_time.txt", "softmax_time.txt")):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(prof, f"{tag}_{dst}"))
    b7 = glob.glob(os.path.join(src, "block7b", "*", "*kernel_stats.csv"))
    if b7:
        shutil.copy(b7[0], os.path.join(prof, f"{tag}_block7b_kernel_stats.csv"))
    paths = {}
    for name in ("fetch", "write"):
        path = glob.glob(os.path.join(src, name, "*", "*counter_collection.csv"))[0]
        paths[name] = path
        rows = list(csv.DictReader(open(path)))
        keep = [r for r in rows if "miq::" in r["Kernel_Name"]]
        with open(os.path.join(prof, f"{tag}_pmc_{name}_size.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=rows[0].keys())
            w.writeheader()
            w.writerows(keep)
    per_kernel = os.path.join(prof, f"{tag}_pmc_traffic_per_kernel.json")
    subprocess.run([sys.executable, os.path.join(REPO, "tools", "pmc_summary.py"), paths["fetch"], paths["write"], per_kernel],
                   check=True, stdout=subprocess.DEVNULL)
    fam = json.load(open(per_kernel))["families_total_bytes"]
    # every pass of the eager run is dispatched and counted (warm-up, the instrumented pass, the leave-one-out sequences,
    # the timed steps); each family runs equally often, the row forward with two launches per pass
    steps = fam["rowquant_fwd"]["dispatches"] / 2
    assert steps == int(steps) and fam["rowquant_bwd"]["dispatches"] == 2 * steps and fam["gather"]["dispatches"] == 2 * steps, fam
    t = {"_per": "step"}
    for k in ("act_bwd", "act_calib_quant_fwd", "rowquant_bwd", "rowquant_fwd"):
        t[k] = round((fam[k]["read"] + fam[k]["write"]) / steps)
    g = round((fam["gather"]["read"] + fam["gather"]["write"]) / steps / 2)
    t["gather_fwd"] = t["gather_bwd"] = g
    t["_source"] = ("HBM bytes per bench step and kernel family = sum over the family's kernels of (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                    "from rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --eager --steps 2 "
                    f"--warmup 1 --no-cpu-baseline --no-block-calib --no-families` (profiles/{tag}_pmc_*.csv); bench.py divides by the family's "
                    "launches per step")
    dig = os.path.join(src, "csrc.digest")
    if os.path.exists(dig):
        t["_csrc_digest"] = open(dig).read().strip()
    json.dump(t, open(os.path.join(prof, "traffic.json"), "w"), indent=1)
    print(json.dumps(t, indent=1))


if __name__ == "__main__":
    main()
