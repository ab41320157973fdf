#!/usr/bin/env python3
"""Per-launch durations and the idle gaps between consecutive kernels of a rocprofv3 kernel trace (csv), for the
activation-quantizer kernels: python3 tools/dev/trace_gaps.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
agg = collections.defaultdict(lambda: [[], []])
prev_end = None
for r in rows:
    n = r["Kernel_Name"]
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if "miq::" in n:
        short = n.split("(")[0].replace("void miq::", "")
        key = (short, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
        agg[key][0].append(en - st)
        if prev_end is not None:
            agg[key][1].append(st - prev_end)
    prev_end = en
for k, (d, g) in sorted(agg.items()):
    d.sort(); g.sort()
    print("%-44s grid %6s %4s %2s wg %4s  n=%3d  dur med %6.1f us   gap before med %5.1f us" % (
        k[0][:44], k[1], k[2], k[3], k[4], len(d), d[len(d) // 2] / 1e3, (g[len(g) // 2] / 1e3) if g else -1))
