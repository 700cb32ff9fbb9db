cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_mixed; rm -rf $O; mkdir -p $O
export AMC_RTC_CACHE_DIR=$O/cache; mkdir -p $AMC_RTC_CACHE_DIR
for v in 0 1; do
  AMC_CLASS_PER_MOVE=$v python3 $R/tools/gpu_workload.py mixed > $O/plain_$v.log 2>&1 && tail -1 $O/plain_$v.log
  AMC_CLASS_PER_MOVE=$v rocprofv3 --kernel-trace --stats -d $O/trace_$v --output-format csv -- python3 $R/tools/gpu_workload.py mixed > $O/prof_$v.log 2>&1 && tail -1 $O/prof_$v.log
  cp $(ls $O/trace_$v/*/*_kernel_stats.csv | head -1) $O/kernel_stats_$v.csv && head -6 $O/kernel_stats_$v.csv | cut -c1-200
done
rm -rf $O/trace_* $O/cache
