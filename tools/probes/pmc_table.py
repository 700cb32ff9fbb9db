#!/usr/bin/env python3
"""Developer tool: per-kernel means of rocprofv3 counter_collection.csv files given on the command line."""
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.1f}  n={len(v)}")
