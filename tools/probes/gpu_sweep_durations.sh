# rocprofv3 kernel-trace averages of the plain K = 2 sweep and fused PGMC launches per package variant, two rounds
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/sweep_durations.txt
: > $O
cd /tmp; export TMPDIR=/tmp
for r in 1 2; do
for v in ${VARIANTS:-base packed planes}; do
    export AMC_PKG_ROOT=$R/tools/_variants/$v
    D=/tmp/sd_${v}
    rocprofv3 --kernel-trace --stats -d $D --output-format csv -- python3 $R/tools/probes/${PROBE:-k2_plain.py} > /dev/null 2>&1
    f=$(find $D -name "*kernel_stats.csv" | head -1)
    python3 - "$f" "$v" >> $O <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "fold_log" in r["Name"] or "sweep_kernel" in r["Name"] or "pg_estimate" in r["Name"]:
        print(f'{sys.argv[2]:10s} {r["Name"][:78]:78s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us')
PY
    rm -rf $D
done
done
cat $O
