#!/usr/bin/env python3
"""Markdown table of SQ counters per kernel from a rocprofv3 --pmc counter_collection CSV
(bench/pmc_step.sh): cycles, MFMA utilisation, wait / issue-stall / active shares, vector
instructions and LDS conflict cycles per dispatch."""
import collections
import csv
import glob
import re
import sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for name, cs in agg.items():
    if "dtf" not in name or "GRBM_GUI_ACTIVE" not in cs:
        continue
    mean = {k: sum(v) / len(v) for k, v in cs.items()}
    n = len(cs["GRBM_GUI_ACTIVE"])
    short = re.sub(r"\(.*", "", name.replace("void dtf::(anonymous namespace)::", "").replace("dtf::(anonymous namespace)::", ""))
    short = short.replace(", ", ",").replace("false", "0").replace("true", "1")
    kcyc = mean["GRBM_GUI_ACTIVE"] / 8 / 1e3
    wave = max(1.0, mean.get("SQ_WAVE_CYCLES", 1.0))
    rows.append((kcyc * n, f"| `{short}` | {n} | {kcyc:.0f} | {100 * mean.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (kcyc * 1e3 * 1024):.0f}% | "
                 f"{100 * mean.get('SQ_WAIT_ANY', 0) / wave:.0f}% | {100 * mean.get('SQ_WAIT_INST_ANY', 0) / wave:.0f}% | "
                 f"{100 * mean.get('SQ_ACTIVE_INST_ANY', 0) / wave:.0f}% | {mean.get('SQ_INSTS_VALU', 0) / 1e6:.2f} | {mean.get('SQ_LDS_BANK_CONFLICT', 0) / 1e6:.2f} |"))
print("| kernel | dispatches | kcycles | MFMA util | wait | issue stall | active | VALU (M) | LDS conflict (M) |")
print("|---|---:|---:|---:|---:|---:|---:|---:|---:|")
for _, line in sorted(rows, reverse=True):
    print(line)
