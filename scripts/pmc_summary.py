#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + PMC passes) of scripts/profile_gpu.sh into one text block."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
for f in sorted(glob.glob(root + "/stats/**/*kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "lz::" in r["Name"]:
            print(f'{r["Name"][:70]:70s} calls={r["Calls"]} avg_ns={float(r["AverageNs"]):.0f} min={r["MinNs"]} max={r["MaxNs"]}')
for f in sorted(glob.glob(root + "/pmc*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "lz::" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(f.split("/")[-2], k, {c: round(sum(x) / len(x), 1) for c, x in v.items()})
