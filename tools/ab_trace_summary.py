#!/usr/bin/env python3
"""Developer tool: per-variant kernel averages out of the kernel_stats.csv files tools/gpu_ab_trace.sh collects."""
import collections, csv, glob, os, re, sys
d = sys.argv[1]
tab = collections.defaultdict(lambda: collections.defaultdict(list))      # (workload, kernel) -> variant -> [avg us per round]
for f in sorted(glob.glob(os.path.join(d, "*_kernel_stats.csv"))):
    m = re.match(r"(.+)_(k2|pgmc|est|ladder_\d+|vec1|vec|mixed)_(\d+)_kernel_stats\.csv", os.path.basename(f))
    if not m:
        continue
    variant, wl = m.group(1), m.group(2)
    for r in csv.DictReader(open(f)):
        name = r["Name"]
        if not re.search(r"sweep_kernel|pg_estimate|fold_log|reduce_kernel|pg_accumulate|pg_update|pg_tail", name):
            continue
        short = re.sub(r"^void amc::", "", name)
        short = re.sub(r"\(.*$", "", short)
        if int(r["Calls"]) < 5:
            continue
        tab[(wl, short)][variant].append(float(r["AverageNs"]) / 1e3)
variants = sorted({v for k in tab for v in tab[k]})
print("average kernel duration, us (one figure per round); variants:", ", ".join(variants))
for (wl, k) in sorted(tab):
    print(f"[{wl}] {k}")
    for v in variants:
        if v in tab[(wl, k)]:
            print(f"    {v:20s} " + "  ".join(f"{x:7.2f}" for x in tab[(wl, k)][v]))
