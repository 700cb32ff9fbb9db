# same-box A/B of the per-chain counter storage (variants under tools/_variants, made with tools/gpu_ab.py snapshot):
#   base    u16 counters for the first 65 535 steps, widened to u32 arrays afterwards; step log one byte per chain
#   packed  the same with the step log at two chains per byte (K <= 4)
#   planes  packed log; counters as two u16 planes, the high plane read (written on carry) once 65 535 steps are counted
# PRECOUNT=70000 starts the count beyond the 16-bit mark: the regime a long run is in.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/counter_planes_ab.txt
: > $O
for r in 1 2 3; do
  for v in base packed planes; do
    for pc in 0 70000; do
      for c in 3 5; do
        AMC_PKG_ROOT=$R/tools/_variants/$v PRECOUNT=$pc STEPS=4000 timeout -k 10 120 python3 $R/tools/gpu_configs.py $c | sed "s/^/$v /" | sed 's/(.*rows deferred)//' | cut -c1-110 >> $O
      done
    done
  done
done
cat $O
