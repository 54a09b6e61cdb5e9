#!/bin/bash
# bytes leaving the L2s per launch of k_conv3x3_c32_bf16 (tools/c32_probe.py under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one
# pass each; corrected as MI355X_MICROARCH.md prescribes by tools/pmc_summary.py).  usage (GPU box): bash tools/pmc_c32.sh
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_c32; mkdir -p $O
python $R/tools/c32_probe.py > $O/timing.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
for pass in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $pass --output-format csv -d $O/$pass -- python3 $R/tools/c32_probe.py 4 > $O/$pass.txt 2>/dev/null
done
cd $R
{ cat $O/timing.txt; python - $O <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_conv3x3_c32_bf16" in k or "k_stem_fwd" in k:
            acc[k[k.index("k_"):].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    # FETCH_SIZE counts 32-byte units of 64 on gfx950 (x 2), both counters are in KB
    fe = 2.0 * sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"])) / 1e3 if "FETCH_SIZE" in d else float("nan")
    wr = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"])) / 1e3 if "WRITE_SIZE" in d else float("nan")
    print("%-52s launches %3d: fetched %.0f MB, written %.0f MB per launch (leaving the L2s)" % (k, len(d.get("FETCH_SIZE", [])), fe, wr))
PY
} > $O/summary.txt
cat $O/summary.txt
find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
