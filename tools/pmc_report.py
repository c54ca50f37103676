#!/usr/bin/env python3
"""Summarise gpurun_out/pmc_<tag>/g*_counter_collection.csv: per kernel, sum each counter over dispatches of the main kernel."""
import csv, glob, sys, collections, os
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "rt_"
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in sorted(glob.glob(os.path.join(d, "*counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if pat not in k:
            continue
        k = k.split("(")[0].replace("void ", "")
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k][row["Counter_Name"]] += 1
for k in tot:
    print(k)
    for c in sorted(tot[k]):
        n = cnt[k][c]
        print(f"   {c:28s} {tot[k][c]/n:18.1f}  (avg over {n} dispatches)")
