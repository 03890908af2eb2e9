#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (counter_collection.csv) per kernel name: sum of each counter over dispatches."""
import csv, glob, sys, collections, os
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "sx_k" not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (f, k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"    {c:28s} {v:.6g}")
