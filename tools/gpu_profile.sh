#!/bin/bash
# Developer tool, run on the GPU box (gpurun -- 'bash tools/gpu_profile.sh'): rocprofv3 kernel-trace summaries and
# PMC passes (own runs, counters only) of bench.py and of the workloads of tools/gpu_workload.py; tools/summarize_profiles.py
# turns what lands in gpurun_out/$TAG/ into the files to commit under profiles/ ($TAG_*; TAG = AMC_ROUND_TAG, default r05).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${AMC_ROUND_TAG:-r06}
export AMC_ROUND_TAG=$TAG
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
# the GPU box has no .git: the caller leaves the commit in .profile_commit (git rev-parse --short HEAD > .profile_commit)
[ -f $R/.profile_commit ] && export AMC_COMMIT=$(cat $R/.profile_commit)
cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err
echo "bench done"
rocprofv3 --kernel-trace --stats -d $O/bench_trace --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-ladder > $O/bench_prof.json 2> $O/bench_prof.err
echo "bench trace done"
W="python3 $R/tools/gpu_workload.py"
export AMC_RTC_CACHE_DIR=$O/rtc_cache; mkdir -p $AMC_RTC_CACHE_DIR      # the script-defined workloads build once, outside the profiled runs
for wl in vec1 vec mixed vec2; do python3 $R/tools/gpu_workload.py $wl > /dev/null 2>&1; done
export PIPELINED=1      # callbacks read one period late, as the host mirror's StoreCallbacks does; the at-once figure is logged next to it
# k2 / pgmc: BASELINE configs 3 / 5 (callbacks every 10 ask for sum e: COLS=1, the callbacks of those configs); vec1 / vec / mixed: the PGMC
# time step of a script-defined one-parameter policy, of the two-parameter drift + width policy, of a two-class pool (round 5)
for wl in "ladder 10000000" "ladder 40000000" "ladder 160000000" "k2" "pgmc" "est" "vec1" "vec" "mixed"; do
  tag=$(echo $wl | tr ' ' '_')
  rocprofv3 --kernel-trace --stats -d $O/$tag/trace --output-format csv -- $W $wl > $O/$tag.log 2>&1
  case $wl in k2|pgmc) PIPELINED=0 $W $wl 2>&1 | tail -1 | sed 's/^/unprofiled, callback read at once: /' >> $O/$tag.log
                       PIPELINED=1 $W $wl 2>&1 | tail -1 | sed 's/^/unprofiled: /' >> $O/$tag.log
                       PRECOUNT=70000 PIPELINED=1 $W $wl 2>&1 | tail -1 | sed 's/^/unprofiled, 70000 steps counted before (the regime after the 16-bit mark): /' >> $O/$tag.log
                       PRECOUNT=70000 rocprofv3 --kernel-trace --stats -d $O/${tag}_late/trace --output-format csv -- $W $wl > /dev/null 2>&1;; esac
  LAUNCHES=120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/$tag/pmc_fetch --output-format csv -- $W $wl > /dev/null 2>&1
  LAUNCHES=120 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/$tag/pmc_write --output-format csv -- $W $wl > /dev/null 2>&1
  LAUNCHES=120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $O/$tag/pmc_sq --output-format csv -- $W $wl > /dev/null 2>&1
  LAUNCHES=120 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --kernel-trace -d $O/$tag/pmc_sq2 --output-format csv -- $W $wl > /dev/null 2>&1
  echo "$tag done: $(cat $O/$tag.log | tail -1)"
done
# kernel traces only: the K = 2 pool of the two-parameter policy with both moves learnable (a chain of launches), the pool of the
# reference's test/pgmc_test.jl (K = 7, six optimisers, q_batch_size = 10)
for wl in vec2 pgmc7; do
  rocprofv3 --kernel-trace --stats -d $O/$wl/trace --output-format csv -- $W $wl > $O/$wl.log 2>&1
  echo "$wl done: $(cat $O/$wl.log | tail -1)"
done
# the per-dispatch traces and counter tables are large (gpurun merges at most 64 MiB back): summarise them here, keep
# the summaries (gpurun_out/${TAG}_out/ -> copied into profiles/ by hand) and drop the raw tables
AMC_PROFILE_OUT=$R/gpurun_out/${TAG}_out python3 $R/tools/summarize_profiles.py
cp $O/*.log $O/bench_n1.err $R/gpurun_out/${TAG}_out/ 2>/dev/null
rm -rf $O
du -sh $R/gpurun_out/${TAG}_out
cat $R/gpurun_out/${TAG}_out/${TAG}_bench_n1.json
