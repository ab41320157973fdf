#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --pmc pass (counter_collection.csv) over tools/row_pmc.py:
    python3 tools/row_pmc_summary.py <counter_collection.csv> <out.json>
For every kernel (template arguments kept, call arguments dropped): launches, and per launch the wave cycles, the share of
them a wave spent issuing any instruction / issuing VALU instructions / parked on s_waitcnt or a barrier / stalled at
issue, and VALU instructions per wave.  SQ cycle counters count quad-cycles; only ratios are reported."""
import collections
import csv
import json
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for r in rows:
        name = r["Kernel_Name"]
        if "miq::" not in name:
            continue
        short = name.split("(")[0].replace("void miq::", "")
        per[short][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[short].add(r["Dispatch_Id"])
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
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    for k, v in out.items():
        print(f"{k[:70]:70s} n={v['launches']:2d} any {v['active_any_frac']:.3f} valu {v['active_valu_frac']:.3f} "
              f"wait {v['wait_any_frac']:.3f} stall {v['wait_inst_any_frac']:.3f} valu/wave {v['valu_insts_per_wave']:.0f}")


if __name__ == "__main__":
    main()
