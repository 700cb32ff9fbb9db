# Developer probe (round 3): effective shader clock of the headline kernel in and out of cache -- GRBM_GUI_ACTIVE (all XCDs) per
# dispatch against the same dispatch's duration in the SAME rocprofv3 run (counter collection + kernel trace only).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/clk; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for m in 10000000 40000000 160000000; do
  LAUNCHES=120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -d $O/m$m --output-format csv -- python3 $R/tools/gpu_workload.py ladder $m > $O/m$m.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for m in (10000000, 40000000, 160000000):
    cf = sorted(glob.glob("$O/m%d/*/*counter_collection.csv" % m))[-1]
    kf = sorted(glob.glob("$O/m%d/*/*kernel_trace.csv" % m))[-1]
    dur = {}
    for r in csv.DictReader(open(kf)):
        if "sweep_kernel<0, false, false, false, true, false>" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    cyc = collections.defaultdict(dict)
    for r in csv.DictReader(open(cf)):
        if "sweep_kernel<0, false, false, false, true, false>" in r["Kernel_Name"]:
            cyc[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = [i for i in dur if i in cyc][len(dur) // 4:]
    g = sum(cyc[i]["GRBM_GUI_ACTIVE"] for i in ids) / len(ids)
    us = sum(dur[i] for i in ids) / len(ids)
    print("M=%d: %d dispatches, %.1f us, GRBM_GUI_ACTIVE/8 = %.0f cycles (%.0f per 1e7 chains) -> %.0f cycles per us" % (m, len(ids), us, g / 8, g / 8 / (m / 1e7), g / 8 / us))
PY
rm -rf $O
