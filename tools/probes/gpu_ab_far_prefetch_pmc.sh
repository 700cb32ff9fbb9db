# Developer tool (round 3): SQ wait counters of the headline kernel at 1.6e8 chains with and without the far prefetch
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/pf_pmc; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for pf in 0 1; do
  AMC_FAR_PREFETCH=$pf LAUNCHES=40 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -d $O/pf$pf --output-format csv -- python3 $R/tools/gpu_workload.py ladder 160000000 > $O/pf$pf.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for pf in (0, 1):
    f = sorted(glob.glob("$O/pf%d/*/*counter_collection.csv" % pf))[-1]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "sweep_kernel<0, false, false, false, true, false>" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v[len(v)//4:]) / len(v[len(v)//4:]) for k, v in agg.items()}
    print("AMC_FAR_PREFETCH=%d at 1.6e8 chains: wait_any/wave_cycles %.3f  wait_inst_any/wave_cycles %.3f  VALUBusy %.3f  launches %d" % (
        pf, m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"],
        m["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (m["GRBM_GUI_ACTIVE"] / 8), len(agg["SQ_WAIT_ANY"])))
PY
rm -rf $O
