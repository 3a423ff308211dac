#!/usr/bin/env python3
"""Aggregates a rocprofv3 --pmc counter_collection CSV per kernel name (mean per dispatch)."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
files = glob.glob(root + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in sorted(agg.items()):
    if "dtf" not in name:
        continue
    short = name.replace("void dtf::(anonymous namespace)::", "").replace("dtf::(anonymous namespace)::", "")[:60]
    parts = [f"{k}={sum(v)/len(v):.4g}" for k, v in sorted(cs.items())]
    n = len(next(iter(cs.values())))
    print(f"{short:60s} n={n:3d} " + " ".join(parts))
