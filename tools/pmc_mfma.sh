#!/bin/bash
# MFMA-pipe occupancy of the 3x3 conv kernels from SQ counters (run on the GPU box): two --pmc passes of one
# split-math fp32 launch shape (tools/conv_probe.py; VD_PMC_MATH = 1 | f16x2 | f16x2nh, default f16x2 = the halo loop) and of the bf16 kernel (tools/stamp_conv.py needs the stamped
# build, so the bf16 pass uses bench.py's detect step instead).
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_mfma; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for pass in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d $O/f32_$tag -- python3 $R/tools/conv_probe.py 256 512 3 1 26 5 ${VD_PMC_MATH:-f16x2} > $O/f32_$tag.txt 2>/dev/null
  rocprofv3 --pmc $pass --output-format csv -d $O/bf16_$tag -- python3 $R/bench.py --mode detect --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline > $O/bf16_$tag.txt 2>/dev/null
done
cd $R
for d in $O/f32_*/ ; do python tools/pmc_summary.py $d k_conv_igemm; done > $O/summary_f32.txt
for d in $O/bf16_*/ ; do python tools/pmc_summary.py $d k_conv_igemm_bf16; done > $O/summary_bf16.txt
cat $O/f32_GRBM_GUI_ACTIVE.txt; cat $O/summary_f32.txt; head -60 $O/summary_bf16.txt
find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
