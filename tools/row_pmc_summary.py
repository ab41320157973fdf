#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --pmc pass (counter_collection.csv) over tools/row_pmc.py:
    python3 tools/row_pmc_summary.py <counter_collection.csv> <out.json>
For every kernel (template arguments kept, call arguments dropped): launches, and per launch the wave cycles, the share of
them a wave spent issuing any instruction / issuing VALU instructions / parked on s_waitcnt or a barrier / stalled at
issue, and VALU instructions per wave.  SQ cycle counters count quad-cycles; only ratios are reported.
Two estimates relate the counters to the machine instead of to the waves: with the dispatch durations of the same file
(End_Timestamp - Start_Timestamp, ns), 1024 SIMDs and a 2.4 GHz clock, `waves_per_simd_est` = wave quad-cycles / SIMD
quad-cycles of the launch and `simd_valu_busy_est` = quad-cycles some wave of a SIMD issued a VALU instruction / SIMD
quad-cycles: the share of the vector units' time that is spent, whatever the number of waves that share it (the clock
under load is lower than 2.4 GHz, so both read a little high; a value near 1 means the kernel is bound by the vector
unit, not by memory latency)."""
import collections
import csv
import json
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    dur = collections.defaultdict(dict)
    for r in rows:
        name = r["Kernel_Name"]
        if "miq::" not in name:
            continue
        short = name.split("(")[0].replace("void miq::", "")
        per[short][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[short].add(r["Dispatch_Id"])
        if r.get("Start_Timestamp") and r.get("End_Timestamp"):
            dur[short][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    out = {}
    for k, c in sorted(per.items()):
        wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        waves = c.get("SQ_WAVES", 0.0) or 1.0
        out[k] = {
            "launches": len(launches[k]),
            "active_any_frac": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4),
            "active_valu_frac": round(c.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 4),
            "wait_any_frac": round(c.get("SQ_WAIT_ANY", 0.0) / wc, 4),
            "wait_inst_any_frac": round(c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4),
            "valu_insts_per_wave": round(c.get("SQ_INSTS_VALU", 0.0) / waves, 1),
            "busy_over_wave_cycles": round(c.get("SQ_BUSY_CYCLES", 0.0) / wc, 4),
        }
        simd_quads = sum(dur[k].values()) * 2.4 / 4 * 1024  # SIMD quad-cycles of all launches of this kernel
        if simd_quads > 0:
            out[k]["waves_per_simd_est"] = round(c.get("SQ_WAVE_CYCLES", 0.0) / simd_quads, 2)
            out[k]["simd_valu_busy_est"] = round(c.get("SQ_ACTIVE_INST_VALU", 0.0) / simd_quads, 2)
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    for k, v in out.items():
        print(f"{k[:70]:70s} n={v['launches']:2d} any {v['active_any_frac']:.3f} valu {v['active_valu_frac']:.3f} "
              f"wait {v['wait_any_frac']:.3f} stall {v['wait_inst_any_frac']:.3f} valu/wave {v['valu_insts_per_wave']:.0f}")


if __name__ == "__main__":
    main()
