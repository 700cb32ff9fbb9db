#!/bin/bash
# Developer probe: does a HIP runtime knob move the single-sweep launch time?  (same box, interleaved)
cd "$(dirname "$0")/.."
run() { env "$@" python3 bench.py --no-cpu-baseline --no-ladder --repeats 3 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s launch %.2f us  step %.2f us  min %.2f  fused %.2f' % ('$*', d['roofline']['avg_launch_us'], d['ms_per_step']*1e3, d['repeat']['ms_per_step_min']*1e3, d['fused_sweepstep16']['us_per_sweep_min']))"; }
for round in 1 2; do
  run X=1
  run HIP_FORCE_DEV_KERNARG=1
  run HIP_FORCE_DEV_KERNARG=0
  run AMD_OPT_FLUSH=0
  run DEBUG_HIP_KERNARG_COPY_OPT=0
  run GPU_FLUSH_ON_EXECUTION=1
done
