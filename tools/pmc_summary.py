#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per kernel family.
usage: pmc_summary.py <dir> [kernel-substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if want and want not in k:
            continue
        short = k.split("(")[0][-60:]
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-34s n=%-4d mean=%.4g" % (c, len(v), sum(v) / len(v)))
