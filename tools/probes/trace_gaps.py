#!/usr/bin/env python3
"""Developer tool: per-kernel mean duration and mean gap to the previous kernel from a rocprofv3 kernel_trace.csv."""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
prev_end = None
for r in rows:
    n = r["Kernel_Name"][:70]; s, t = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[n].append(t - s)
    if prev_end is not None: gap[n].append(s - prev_end)
    prev_end = t
for n in dur:
    d, g = dur[n], gap[n] or [0]
    d2 = sorted(d); g2 = sorted(g)
    print(f"{n:72s} n={len(d):6d} dur mean {sum(d)/len(d)/1e3:8.2f} med {d2[len(d2)//2]/1e3:8.2f} us | gap-before med {g2[len(g2)//2]/1e3:7.2f} mean {sum(g)/len(g)/1e3:7.2f} us")
