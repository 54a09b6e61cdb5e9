#!/bin/bash
# MFMA-pipe occupancy, LDS conflicts and L2-side traffic of the two weight-gradient kernels on one 3x3 layer shape
# (run on the GPU box): --pmc passes of tools/wgrad_probe.py, generic kernel (f16x2) against the halo ring (f16x2h).
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_wgrad; mkdir -p $O
SHAPE=${VD_PMC_SHAPE:-"256 512 26"}
cd /tmp && export TMPDIR=/tmp
for mode in f16x2 f16x2h; do
  for pass in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
    tag=$(echo $pass | cut -d' ' -f1)
    rocprofv3 --pmc $pass --output-format csv -d $O/${mode}_$tag -- python3 $R/tools/wgrad_probe.py $SHAPE $mode > $O/${mode}_$tag.txt 2>/dev/null
  done
done
cd $R
for mode in f16x2 f16x2h; do
  echo "== $mode  ($(cat $O/${mode}_GRBM_GUI_ACTIVE.txt | tail -1))"
  for d in $O/${mode}_*/ ; do python tools/pmc_summary.py $d k_conv_wgrad; done
done > $O/summary.txt
cat $O/summary.txt
find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
