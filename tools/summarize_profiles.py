#!/usr/bin/env python3
"""Turns the rocprofv3 outputs merged into gpurun_out/ into the committed summaries under profiles/.

  profiles/<tag>_kernel_stats.csv    rocprofv3 --kernel-trace --stats summary of `bench.py`
  profiles/<tag>_pmc_summary.json    per-launch averages of the PMC passes for the sweep kernel
  profiles/pmc_traffic.json          HBM bytes per sweep launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE),
                                     read by bench.py for roofline.traffic
"""
import csv, glob, json, os, shutil, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = os.path.join(ROOT, "profiles"); os.makedirs(out, exist_ok=True)
g = os.path.join(ROOT, "gpurun_out")
def newest(pattern):
    """gpurun merges every call's files into gpurun_out/: take the latest run's."""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:]


for f in newest(os.path.join(g, "prof", "*", "*_kernel_stats.csv")):
    shutil.copy(f, os.path.join(out, f"{tag}_kernel_stats.csv"))
    print(open(f).read()[:900])
summary = collections.OrderedDict()
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    for f in newest(os.path.join(g, d, "*", "*counter_collection.csv")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "sweep_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = (r["Kernel_Name"], r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"])
        for k, v in agg.items():
            summary[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
        if agg:
            summary["_kernel"] = dict(zip(("name", "grid", "workgroup", "vgpr", "sgpr", "lds"), meta))
if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
    fetch_kb, write_kb = summary["FETCH_SIZE"]["mean_per_launch"], summary["WRITE_SIZE"]["mean_per_launch"]
    traffic = (2.0 * fetch_kb + write_kb) * 1024.0
    summary["_traffic"] = {"fetch_bytes_corrected_x2": 2 * fetch_kb * 1024, "write_bytes": write_kb * 1024,
                           "hbm_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": 16 * 10_000_000,
                           "note": "FETCH_SIZE counts 64 B per 128-B request on gfx950 (MI355X_MICROARCH.md, HBM): doubled"}
    extra = {}
    if "SQ_ACTIVE_INST_VALU" in summary and "GRBM_GUI_ACTIVE" in summary:
        # rocprof's VALUBusy = SQ_ACTIVE_INST_VALU * 4 / SIMDs / GRBM_GUI_ACTIVE; rocprofv3 reports GRBM_GUI_ACTIVE summed over
        # the 8 XCDs (MI355X_MICROARCH.md, DVFS note), so the per-XCD busy cycles are 1/8 of it.  256 CUs x 4 SIMDs.
        busy = summary["SQ_ACTIVE_INST_VALU"]["mean_per_launch"] * 4.0 / 1024.0 / (summary["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0)
        summary["_valu_busy"] = {"frac": busy, "formula": "SQ_ACTIVE_INST_VALU * 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)"}
        extra["sweep_kernel_valu_busy"] = busy
    json.dump({"sweep_kernel_bytes_per_launch": traffic, "source": f"profiles/{tag}_pmc_summary.json", **extra},
              open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
json.dump(summary, open(os.path.join(out, f"{tag}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
