set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${VD_PROFILE_TAG:-r01j}; mkdir -p $O
cd $R
# every bench of this run reads / extends ONE tuning table: copied into viddet_amd/tune/ afterwards (tools/collect_profiles.sh)
# (VD_PROFILE_KEEP_TABLE=1: run on the committed table under viddet_amd/tune/ instead - the kernels the driver's run takes)
if [ "${VD_PROFILE_KEEP_TABLE:-0}" != "1" ]; then export VD_TUNE_CACHE=$O/tune_table.json; fi
python bench.py > $O/train_bench.json 2> $O/train_bench.err
echo "train done"; cut -c1-160 $O/train_bench.json
python bench.py --mode detect --no-cpu-baseline > $O/detect_bench.json 2> $O/detect.err
python bench.py --mode detect --dtype bf16 --no-cpu-baseline > $O/detect_bf16_bench.json 2> $O/detect_bf16.err
echo "detect done"; cut -c1-160 $O/detect_bench.json; cut -c1-160 $O/detect_bf16_bench.json
python bench.py --window 3 --batch 16 --classes 30 --no-cpu-baseline > $O/train_k3_bench.json 2> $O/k3.err
python bench.py --dtype bf16 --no-cpu-baseline > $O/train_bf16products_bench.json 2> $O/bf16p.err
echo "k3/bf16p done"; cut -c1-160 $O/train_k3_bench.json; cut -c1-160 $O/train_bf16products_bench.json
python bench.py --storage bf16 --no-cpu-baseline > $O/train_bf16storage_bench.json 2> $O/bf16s.err
python bench.py --storage bf16 --size 608 --classes 285 --batch 32 --no-cpu-baseline > $O/train_cfg4_bf16storage_bench.json 2> $O/bf16s4.err
python bench.py --dtype bf16 --size 608 --classes 285 --batch 32 --no-cpu-baseline > $O/train_cfg4_bf16products_bench.json 2> $O/bf16p4.err
echo "bf16 storage done"; cut -c1-160 $O/train_bf16storage_bench.json; cut -c1-160 $O/train_cfg4_bf16storage_bench.json; cut -c1-160 $O/train_cfg4_bf16products_bench.json
# second start of the headline bench on the finished table: nothing is timed again (start-up time in the .time file)
S0=$SECONDS
python bench.py --no-cpu-baseline --no-native --no-detect --steps 5 --warmup 2 > $O/train_second_start.json 2> $O/second.err
echo "$((SECONDS - S0)) s wall for a whole bench.py process (import, plan build from the persisted table, 2 + 5 steps, roofline replay)" > $O/second_start.time
cat $O/second_start.time
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-native --no-detect > $O/prof_train.json 2> $O/prof_train.err
echo "prof train done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_detect_bf16 -- python3 $R/bench.py --mode detect --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_detect_bf16.json 2> $O/prof_detect_bf16.err
python $R/tools/last_step_kernels.py $O/prof_detect_bf16 $O/detect_bf16_last_step_kernels.json --detect
echo "prof detect done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16s -- python3 $R/bench.py --storage bf16 --size 608 --classes 285 --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_bf16s.json 2> $O/prof_bf16s.err
python $R/tools/last_step_kernels.py $O/prof_bf16s $O/train_cfg4_bf16storage_last_step_kernels.json
echo "prof bf16 storage done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-native --no-detect > $O/pmc_fetch.json 2> $O/pmc_fetch.err
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-native --no-detect > $O/pmc_write.json 2> $O/pmc_write.err
echo "pmc write done"
cd $R
# keep only the small summaries
python tools/last_step_kernels.py $O/prof_train $O/train_last_step_kernels.json
# launches per step of the three conv kernels, from the trace of the same command (the autotuner decides how many weight
# gradients take the halo-ring kernel)
LAST=$(python -c "import json,sys; k=json.load(open('$O/train_last_step_kernels.json'))['kernels']; print(','.join('%s=%d' % (n, k[n]['launches']) for n in ('k_conv_igemm','k_conv_wgrad','k_conv_wgrad_halo') if n in k))")
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json --last $LAST | tail -4
# the same step with the weight-gradient GEMMs on the main stream: every launch runs alone, so its rocprof duration is
# comparable with bench.py's per-launch event timing (roofline.avg_launch_ms)
(cd /tmp && VD_OVERLAP=0 rocprofv3 --kernel-trace --output-format csv -d $O/prof_train_serial -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-native --no-detect > $O/prof_train_serial.json 2> $O/prof_train_serial.err)
python tools/last_step_kernels.py $O/prof_train_serial $O/train_last_step_kernels_serial.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
