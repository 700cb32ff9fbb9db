#!/bin/bash
# Developer tool: one workload of tools/gpu_workload.py under several environments on one box, kernel-trace averages, ROUNDS times interleaved.
# usage (GPU box): WL="vec vec1" ENVS="AMC_NO_UNIFORM_DIV=1|AMC_UDIV_MODE=inline|X=1" ROUNDS=2 bash tools/gpu_env_ab.sh <tag>
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-env_ab}; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
export AMC_RTC_CACHE_DIR=$O/cache; mkdir -p $AMC_RTC_CACHE_DIR
IFS='|' read -ra ES <<< "$ENVS"
for round in $(seq 1 ${ROUNDS:-2}); do
  for wl in $WL; do
    for i in "${!ES[@]}"; do
      e="${ES[$i]}"
      env $e timeout -k 5 200 python3 $R/tools/gpu_workload.py $wl > /dev/null 2>&1      # compile outside the profiler
      d=$O/raw_${wl}_${i}_$round
      ( export $e; timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $d --output-format csv -- python3 $R/tools/gpu_workload.py $wl > $O/${wl}_${i}_$round.log 2>&1 ) || { echo "FAILED $wl $e"; tail -5 $O/${wl}_${i}_$round.log; exit 1; }
      f=$(ls $d/*/*_kernel_stats.csv | head -1)
      echo "$wl [$e] round $round: $(grep -h 'us per' $O/${wl}_${i}_$round.log | cut -c1-40) | kernel avg ns: $(grep -E 'pg_estimate_kernel|sweep_kernel' $f | head -1 | awk -F'","|",' '{print $4}' | cut -d, -f3)"
    done
  done
done
rm -rf $O/raw_* $O/cache
