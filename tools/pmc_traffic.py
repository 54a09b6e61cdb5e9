#!/usr/bin/env python
"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE in one, WRITE_SIZE in the other; they do not
fit one pass on gfx950).  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950; both counters are in KB.
Only the LAST n dispatches of a kernel family are used when --last family=n is given (the launches of the final step,
not the plan-time autotuner's trial launches).
usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> [--last k_conv_igemm=164,k_conv_wgrad=75]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

FAMILIES = ["k_conv_igemm_bf16", "k_conv_igemm", "k_conv_wgrad_halo", "k_conv_wgrad", "k_bn_partial", "k_bn_apply_leaky", "k_bn_bwd_apply",
            "k_reduce_slabs", "k_bn_sum_partials", "k_stem_im2col", "k_sgd", "k_yolo_loss", "k_upcat"]


def family(name):
    for f in FAMILIES:
        if f in name:
            return f
    return None


def load(root, counter):
    rows = defaultdict(list)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            fam = family(r["Kernel_Name"])
            if fam:
                rows[fam].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return {k: [v for _, v in sorted(vs)] for k, vs in rows.items()}


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    last = {}
    if "--last" in sys.argv:
        for kv in sys.argv[sys.argv.index("--last") + 1].split(","):
            k, v = kv.split("=")
            last[k] = int(v)
    fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    res = {}
    for fam in FAMILIES:
        if fam not in fe or fam not in wr:
            continue
        f, w = fe[fam], wr[fam]
        n = last.get(fam)
        if n:
            f, w = f[-n:], w[-n:]
        fk, wk = sum(f) / len(f), sum(w) / len(w)
        res[fam] = {"launches": len(f), "fetch_kb_raw": fk, "write_kb": wk,
                    "hbm_mb_corrected": (2.0 * fk + wk) * 1024.0 / 1e6,
                    "scope": ("last %d dispatches (final step)" % n) if n else "all dispatches incl. autotuner trials"}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        print("%-20s n=%-5d HBM %.1f MB/launch" % (k, v["launches"], v["hbm_mb_corrected"]))


if __name__ == "__main__":
    main()
