#!/bin/bash
# copy the judged summaries of one refresh_profiles.sh run from gpurun_out/<tag>/ into profiles/ (tracked): tools/collect_profiles.sh r02a
set -e
T=$1; O=gpurun_out/$T; P=profiles
cp $O/train_bench.json $P/${T}_train_b64_416_bench.json
cp $O/detect_bench.json $P/${T}_detect_b32_608_bench.json
cp $O/detect_bf16_bench.json $P/${T}_detect_b32_608_bf16_bench.json
cp $O/train_k3_bench.json $P/${T}_train_k3_b16_416_bench.json
cp $O/train_bf16products_bench.json $P/${T}_train_b64_416_bf16products_bench.json
cp $O/train_bf16storage_bench.json $P/${T}_train_b64_416_bf16storage_bench.json
cp $O/train_cfg4_bf16storage_bench.json $P/${T}_train_cfg4_b32_608_c285_bf16storage_bench.json
cp $O/train_cfg4_bf16products_bench.json $P/${T}_train_cfg4_b32_608_c285_bf16products_bench.json
cp $O/train_cfg4_bf16storage_last_step_kernels.json $P/${T}_train_cfg4_bf16storage_last_step_kernels.json
cp $O/second_start.time $P/${T}_second_start_seconds.txt
# the tuning table of this library build (keyed by the kernel-source hash): read by every later start
[ -f $O/tune_table.json ] && mkdir -p viddet_amd/tune && python - <<PY
import json, shutil
from viddet_amd.model import TuneCache
doc = json.load(open("$O/tune_table.json"))
assert doc["library"] == TuneCache.library_id(), "the table was timed on other kernel sources"
shutil.copy("$O/tune_table.json", "viddet_amd/tune/gfx950_%s.json" % doc["library"])
print("tuning table:", len(doc["entries"]), "entries ->", "viddet_amd/tune/gfx950_%s.json" % doc["library"])
PY
cp $(ls $O/prof_train/*/*_kernel_stats.csv | head -1) $P/${T}_train_b64_416_kernel_stats.csv
cp $(ls $O/prof_detect_bf16/*/*_kernel_stats.csv | head -1) $P/${T}_detect_b32_608_bf16_kernel_stats.csv
python tools/summarize_rocprof.py $O/prof_train $P/${T}_train_b64_416_kernel_summary.md
python tools/summarize_rocprof.py $O/prof_detect_bf16 $P/${T}_detect_b32_608_bf16_kernel_summary.md
cp $O/pmc_traffic.json $P/${T}_train_b64_416_pmc_traffic.json
cp $O/train_last_step_kernels.json $P/${T}_train_last_step_kernels.json
[ -f $O/detect_bf16_last_step_kernels.json ] && cp $O/detect_bf16_last_step_kernels.json $P/${T}_detect_b32_608_bf16_last_step_kernels.json
cp $O/train_last_step_kernels_serial.json $P/${T}_train_last_step_kernels_serial.json
ls -la $P/${T}_*
