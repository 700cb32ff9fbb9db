# rocprofv3 kernel-trace averages of the ratio fold (callback_acceptance's launch) per counter-storage variant and regime
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/fold_durations.txt
: > $O
cd /tmp; export TMPDIR=/tmp
export STEPS=2000
for v in base packed planes; do
  for pc in 0 70000; do
    export AMC_PKG_ROOT=$R/tools/_variants/$v PRECOUNT=$pc
    D=/tmp/fd_${v}_$pc
    rocprofv3 --kernel-trace --stats -d $D --output-format csv -- python3 $R/tools/gpu_configs.py 3 > /dev/null 2>&1
    f=$(find $D -name "*kernel_stats.csv" | head -1)
    python3 - "$f" "$v precount=$pc" >> $O <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "fold_log" in r["Name"] or "sweep_kernel" in r["Name"]:
        print(f'{sys.argv[2]:24s} {r["Name"][:75]:75s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us')
PY
    rm -rf $D
  done
done
cat $O
