#!/usr/bin/env python
"""Per-family launch count / average duration over the LAST training step of a rocprofv3 --kernel-trace run (the step
between the last two optimiser launches), i.e. without the plan-time autotuner's trial launches that the --stats summary
of the whole process includes.  usage: last_step_kernels.py <dir with *_kernel_trace.csv> <out.json> [--detect]"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def family(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else re.sub(r"<.*", "", name)[:60]


def main():
    f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    if len(sys.argv) > 3 and sys.argv[3] == "--detect":
        # inference: a step ends with its k_nms launch; the last step runs at the calibrated objectness bias (the sweep that
        # finds it - all-pass biases included - comes first and must not be averaged in)
        nms = [i for i, r in enumerate(rows) if "k_nms" in r[2]]
        a, b = nms[-2] + 1, nms[-1] + 1
    else:
        sgd = [i for i, r in enumerate(rows) if "k_sgd" in r[2]]
        # two k_sgd launches per step (weights; gamma/beta/bias): the step is everything after the previous step's second one
        a, b = sgd[-3] + 1, sgd[-1] + 1
    step = rows[a:b]
    agg = defaultdict(lambda: [0, 0])
    for s, e, n in step:
        v = agg[family(n)]
        v[0] += 1
        v[1] += e - s
    out = {"step_wall_ms": round((max(e for _, e, _ in step) - step[0][0]) / 1e6, 3),
           "kernels": {k: {"launches": v[0], "avg_us": round(v[1] / v[0] / 1e3, 2), "total_ms": round(v[1] / 1e6, 3)}
                       for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}}
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    for k in ("k_conv_igemm", "k_conv_wgrad"):
        if k in out["kernels"]:
            print(k, out["kernels"][k])
    print("step wall", out["step_wall_ms"], "ms")


if __name__ == "__main__":
    main()
