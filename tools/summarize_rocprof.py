#!/usr/bin/env python
"""Condense a rocprofv3 `--kernel-trace --stats --output-format csv` run into a per-kernel-family table.
usage: summarize_rocprof.py <dir with *_kernel_stats.csv> [out.md]"""
import csv
import glob
import re
import sys


def family(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    if m:
        return m.group(1)
    return re.sub(r"<.*", "", name)[:60]


def main():
    d = sys.argv[1]
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    fam = {}
    for r in csv.DictReader(open(f)):
        k = family(r["Name"])
        v = fam.setdefault(k, [0, 0.0])
        v[0] += int(r["Calls"])
        v[1] += float(r["TotalDurationNs"])
    tot = sum(v[1] for v in fam.values())
    lines = ["| kernel family | calls | total ms | avg us | % |", "|---|---:|---:|---:|---:|"]
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        lines.append("| %s | %d | %.2f | %.1f | %.1f |" % (k, v[0], v[1] / 1e6, v[1] / v[0] / 1e3, 100 * v[1] / tot))
    lines.append("| **total** | | %.2f | | |" % (tot / 1e6))
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
