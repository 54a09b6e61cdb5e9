#!/usr/bin/env python
"""Achieved HBM rate of the memory-bound kernels: bytes per launch from the PMC passes (tools/pmc_traffic.py JSON:
2 x FETCH_SIZE + WRITE_SIZE) over the average launch duration of a rocprofv3 --kernel-trace --stats run of the same
command.  usage: hbm_table.py <pmc_traffic.json> <kernel_stats.csv> [peak_TBps=8.0]"""
import csv
import json
import sys

pmc = json.load(open(sys.argv[1]))
peak = float(sys.argv[3]) if len(sys.argv) > 3 else 8.0
dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    name = r["Name"]
    for fam in pmc:
        if fam in name and not (fam == "k_conv_igemm" and "bf16" in name):
            d = dur.setdefault(fam, [0.0, 0])
            d[0] += float(r["TotalDurationNs"])
            d[1] += int(r["Calls"])
print("| kernel | launches (stats run) | avg us | MB / launch (PMC) | achieved TB/s | of %.0f TB/s |" % peak)
print("|---|---:|---:|---:|---:|---:|")
for fam, v in pmc.items():
    if fam not in dur or fam.startswith("k_conv"):
        continue
    us = dur[fam][0] / dur[fam][1] / 1e3
    tb = v["hbm_mb_corrected"] * 1e6 / (us * 1e-6) / 1e12
    print("| `%s` | %d | %.1f | %.1f | %.2f | %.2f |" % (fam, dur[fam][1], us, v["hbm_mb_corrected"], tb, tb / peak))
