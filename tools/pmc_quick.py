#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter CSVs per kernel (development aid).
   python tools/pmc_quick.py <rocprof output dir> [kernel name part]"""
import collections
import csv
import glob
import sys

root, part = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "fir_f64")
for f in sorted(glob.glob(root + "/**/*_counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    span = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if part not in k:
            continue
        k = k.split("(")[0][-60:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
        span[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k in agg:
        n = len(disp[k])
        ns = sum(span[k].values()) / n
        print(f"{k}  launches={n}  avg {ns / 1e6:.3f} ms")
        for c, v in sorted(agg[k].items()):
            print(f"    {c:34s} {v / n:16.1f} per launch")
