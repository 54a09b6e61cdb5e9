#!/usr/bin/env python
"""Idle time inside the LAST training step of a rocprofv3 --kernel-trace run: the union of all kernels' [start, end] intervals
against the step's wall time, split at the loss kernel into the forward and backward halves, and the gaps by the kernel
that follows them.  usage: step_gaps.py <dir with *_kernel_trace.csv> [out.json]"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def fam(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name[:40]


def main():
    f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
    sgd = [i for i, r in enumerate(rows) if "k_sgd" in r[2]]
    step = rows[sgd[-3] + 1:sgd[-1] + 1]
    t0, t1 = step[0][0], max(e for _, e, _ in step)
    loss_t = [s for s, e, n in step if "k_yolo_loss" in n][0]
    busy_end, idle, gaps = t0, {"fwd": 0, "bwd": 0}, defaultdict(lambda: [0, 0])
    for s, e, n in step:
        if s > busy_end:                                  # nothing was running between busy_end and s
            half = "fwd" if s <= loss_t else "bwd"
            idle[half] += s - busy_end
            g = gaps[fam(n)]
            g[0] += 1
            g[1] += s - busy_end
        busy_end = max(busy_end, e)
    out = {"step_wall_ms": round((t1 - t0) / 1e6, 3), "forward_wall_ms": round((loss_t - t0) / 1e6, 3),
           "idle_ms": {k: round(v / 1e6, 3) for k, v in idle.items()}, "kernels_in_step": len(step),
           "idle_before_kernel": {k: {"gaps": v[0], "total_us": round(v[1] / 1e3, 1), "avg_us": round(v[1] / v[0] / 1e3, 2)}
                                  for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:12]}}
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
