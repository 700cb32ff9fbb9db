#!/bin/bash
# Developer tool: kernel-trace averages of a list of tools/gpu_workload.py workloads on one box, ROUNDS times interleaved.
# usage (on the GPU box): WL="vec vec_auto vec1 vec1_auto" ROUNDS=2 bash tools/gpu_wl_ab.sh <tag>
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-wl_ab}; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
export AMC_RTC_CACHE_DIR=$O/cache; mkdir -p $AMC_RTC_CACHE_DIR
for round in $(seq 1 ${ROUNDS:-2}); do
  for wl in ${WL:-vec vec_auto}; do
    timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/raw_${wl}_$round --output-format csv -- python3 $R/tools/gpu_workload.py $wl > $O/${wl}_$round.log 2>&1 || { echo "FAILED $wl"; tail -5 $O/${wl}_$round.log; exit 1; }
    f=$(ls $O/raw_${wl}_$round/*/*_kernel_stats.csv | head -1)
    cp $f $O/${wl}_${round}_kernel_stats.csv
    echo "$wl round $round: $(tail -1 $O/${wl}_$round.log | cut -c1-90)"
    head -3 $f | tail -2 | cut -d, -f1-4 | cut -c1-150
  done
done
rm -rf $O/raw_* $O/cache
