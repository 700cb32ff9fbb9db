# same-box A/B: step log with two chains per byte (K <= 4) against one byte per chain
# (variants under tools/_variants: `base`, `packed`, made with tools/gpu_ab.py snapshot)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/packed_ab.txt
: > $O
python3 $R/tools/gpu_ab.py run 3 >> $O
for r in 1 2 3; do
  for v in base packed; do
    for w in 0 1; do
      for c in 3 5; do
        AMC_PKG_ROOT=$R/tools/_variants/$v AMC_WIDE_COUNTERS=$w STEPS=4000 timeout -k 10 120 python3 $R/tools/gpu_configs.py $c | sed "s/^/$v wide=$w /" | cut -c1-150 >> $O
      done
    done
  done
done
cat $O
