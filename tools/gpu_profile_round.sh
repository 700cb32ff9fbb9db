#!/bin/bash
# Developer tool, run on the GPU box (gpurun -- 'bash tools/gpu_profile_round.sh'): the default bench line, the
# rocprofv3 kernel-trace summary of the same command and the PMC passes (own runs, counters only) whose summaries
# tools/summarize_profiles.py then copies into profiles/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O; rm -rf $O/prof $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_sq2
cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err
rocprofv3 --kernel-trace --stats -d $O/prof --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
B="python3 $R/bench.py --no-cpu-baseline --steps 400 --warmup 50 --spinup-s 0.05"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_sq --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT --kernel-trace -d $O/pmc_sq2 --output-format csv -- $B > /dev/null 2>&1
cat $O/bench_n1.json
