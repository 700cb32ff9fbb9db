#!/bin/bash
# Developer tool, run on the GPU box (gpurun -- 'bash tools/gpu_ab_trace.sh [workloads]'): same-box A/B of library variants at
# KERNEL level -- every variant under tools/_variants/ (tools/gpu_ab.py snapshot NAME) runs the workloads of tools/gpu_workload.py
# under rocprofv3 --kernel-trace --stats, interleaved, ROUNDS times; gpurun_out/ab_trace/summary.txt lists the average duration
# of the path's kernels per variant.  (Boxes of the pool differ by several percent: only figures of one call compare.)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/ab_trace
WLS=${1:-"k2 pgmc"}
ROUNDS=${ROUNDS:-2}
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export PIPELINED=1 LAUNCHES=${LAUNCHES:-1500}
for round in $(seq 1 $ROUNDS); do
  for v in $(ls $R/tools/_variants | sort); do
    for wl in $WLS; do
      export AMC_PKG_ROOT=$R/tools/_variants/$v
      unset AMC_BLOCKS_PER_CU AMC_BLOCKS_PER_CU_REDUCE COLS AMC_NO_DEFERRED_UPDATE AMC_RTC_LICM; export COLS=1; [ -f $AMC_PKG_ROOT/env ] && source $AMC_PKG_ROOT/env
      timeout -k 5 150 rocprofv3 --kernel-trace --stats -d $O/raw/${v}_${wl}_$round --output-format csv -- python3 $R/tools/gpu_workload.py $wl > $O/${v}_${wl}_$round.log 2>&1 || { echo "FAILED $v $wl"; tail -5 $O/${v}_${wl}_$round.log; exit 1; }
      f=$(ls -t $O/raw/${v}_${wl}_$round/*/*_kernel_stats.csv | head -1)
      cp $f $O/${v}_${wl}_${round}_kernel_stats.csv
      echo "$v $wl round $round: $(tail -1 $O/${v}_${wl}_$round.log)"
    done
  done
done
rm -rf $O/raw
python3 $R/tools/ab_trace_summary.py $O | tee $O/summary.txt
