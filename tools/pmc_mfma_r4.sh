#!/bin/bash
# Round 4: matrix-pipe occupancy, LDS conflicts and L2-side traffic of k_conv_igemm on 256 -> 512 @26 (batch 64, tile 5, fp16
# split, halo loop): the one-tile-per-workgroup launch against its persistent stream-K form (VD_PROBE_STREAMK=1).
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_mfma_r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for sk in 0 1; do
  for pass in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
    tag=$(echo $pass | cut -d' ' -f1)
    VD_PROBE_STREAMK=$sk rocprofv3 --pmc $pass --output-format csv -d $O/sk${sk}_$tag -- python3 $R/tools/conv_probe.py 256 512 3 1 26 5 f16x2 > $O/sk${sk}_$tag.txt 2>/dev/null
  done
done
cd $R
for sk in 0 1; do
  echo "== streamk=$sk  ($(tail -1 $O/sk${sk}_GRBM_GUI_ACTIVE.txt))"
  for d in $O/sk${sk}_*/ ; do python tools/pmc_summary.py $d k_conv_igemm; done
done > $O/summary.txt
cat $O/summary.txt
find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
