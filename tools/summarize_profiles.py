#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of tools/gpu_profile.sh (merged into gpurun_out/<tag>/; tag = AMC_ROUND_TAG, default r03) into the committed summaries:

  profiles/<tag>_bench_kernel_stats.csv            rocprofv3 --kernel-trace --stats of `python3 bench.py` (the headline command)
  profiles/<tag>_<workload>_kernel_stats.csv       the same for the ladder sizes, the K = 2 sweep (config 3 shape), the fused
                                                 PGMC step (config 5) and the estimator launch alone
  profiles/<tag>_pmc_summary.json                  per workload and kernel: per-launch means of the PMC passes and what follows
                                                 from them (HBM bytes with the gfx950 FETCH_SIZE x2 correction, VALUBusy, VALU
                                                 instructions per wave, wait fractions, LDS bank-conflict cycles)
  profiles/pmc_traffic.json                      headline kernel: bytes per launch + VALUBusy, with commit and kernel-source hash
                                                 (bench.py quotes it as a static figure with that provenance)
  profiles/<tag>_bench_n1.json                     the bench line of the same box
"""
import collections, csv, glob, hashlib, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = os.environ.get("AMC_ROUND_TAG", "r06")
G = os.path.join(ROOT, "gpurun_out", TAG)
OUT = os.environ.get("AMC_PROFILE_OUT", os.path.join(ROOT, "profiles"))       # on the GPU box: a directory under gpurun_out/
os.makedirs(OUT, exist_ok=True)
ALGO_BYTES = {"ladder_10000000": 16 * 10_000_000, "ladder_40000000": 16 * 40_000_000, "ladder_160000000": 16 * 160_000_000,
              "k2": 165_000_000, "pgmc": 165_000_000, "est": 16 * 10_000_000,     # 16 B of state + half a step-log byte per update
              "vec1": 165_000_000, "vec": 165_000_000, "mixed": 165_000_000}
MAIN = {"ladder": "sweep_kernel<0, false, 0, false, true, 0>", "k2": "sweep_kernel<1, true, 1, false, true",
        "pgmc": "pg_estimate_kernel<0, 1, false, 2, 0, false>", "est": "pg_estimate_kernel<0, 1, false, 0, 0, false>",
        # hiprtc forms (POT_CUSTOM = 2 names the script-defined family): the fused time step of the one- / two-parameter script
        # policy, the per-move estimator launch of the two-class pool
        "vec1": "pg_estimate_kernel<2, 1, false, 1, 0, false>", "vec": "pg_estimate_kernel<2, 1, false, 1, 0, false>",
        # round 6: the class pool's time step is ONE launch (every learnable move, the sweep in front)
        "mixed": "pg_estimate_kernel<2, 2, false, 2, 0, false>"}


def one(pattern):
    f = sorted(glob.glob(pattern), key=os.path.getmtime)
    return f[-1] if f else None


def kernel_hash():
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "montecarlo_amd", "csrc")
    for fn in sorted(f for f in os.listdir(csrc) if f.startswith("amc_") and f.endswith(".h") and not f.endswith(".gen.h")):      # every kernel source
        h.update(open(os.path.join(csrc, fn), "rb").read())
    return h.hexdigest()[:16]


summary = collections.OrderedDict()
f = one(os.path.join(G, "bench_trace", "*", "*_kernel_stats.csv"))
if f:
    shutil.copy(f, os.path.join(OUT, f"{TAG}_bench_kernel_stats.csv"))
if os.path.exists(os.path.join(G, "bench_n1.json")):
    shutil.copy(os.path.join(G, "bench_n1.json"), os.path.join(OUT, f"{TAG}_bench_n1.json"))
for wl in ALGO_BYTES:
    f = one(os.path.join(G, wl, "trace", "*", "*_kernel_stats.csv"))
    if not f:
        continue
    shutil.copy(f, os.path.join(OUT, f"{TAG}_{wl}_kernel_stats.csv"))
    late = one(os.path.join(G, wl + "_late", "trace", "*", "*_kernel_stats.csv"))      # the same workload past the 16-bit mark
    if late:
        shutil.copy(late, os.path.join(OUT, f"{TAG}_{wl}_late_kernel_stats.csv"))
    main = MAIN[wl.split("_")[0]]
    entry = collections.OrderedDict()
    # every kernel of the path the workload launched at least five times: name -> calls, average / minimum us (the forms that
    # also leave the callback sums, the folds of the step log, the small single-thread launches)
    entry["kernels"] = collections.OrderedDict(
        (r["Name"].replace("void amc::", "").split("(")[0], dict(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3, min_us=float(r["MinNs"]) / 1e3))
        for r in csv.DictReader(open(f)) if "amc::" in r["Name"] and int(r["Calls"]) >= 5)
    rows = [r for r in csv.DictReader(open(f)) if main in r["Name"]]
    if rows:
        # (the K = 2 workload launches two forms of the sweep kernel: plain, and with the callback sums every 10th step)
        entry["kernel"] = [r["Name"] for r in rows] if len(rows) > 1 else rows[0]["Name"]
        entry["launches_traced"] = sum(int(r["Calls"]) for r in rows)
        entry["avg_us_kernel_trace"] = sum(float(r["TotalDurationNs"]) for r in rows) / entry["launches_traced"] / 1e3
        entry["min_us_kernel_trace"] = min(float(r["MinNs"]) for r in rows) / 1e3
    counters = {}
    meta = None
    for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
        cf = one(os.path.join(G, wl, d, "*", "*counter_collection.csv"))
        if not cf:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(cf)):
            if main in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = dict(grid=int(r["Grid_Size"]), workgroup=int(r["Workgroup_Size"]), vgpr=int(r["VGPR_Count"]),
                            sgpr=int(r["SGPR_Count"]), lds=int(r["LDS_Block_Size"]))
        for k, v in agg.items():
            v = v[len(v) // 4:]                               # skip the first quarter: clock ramp / first-touch launches
            counters[k] = sum(v) / len(v)
    entry["launch"] = meta
    entry["counters_mean_per_launch"] = counters
    d = collections.OrderedDict()
    if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        d["hbm_bytes_per_launch"] = (2.0 * counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024.0
        d["fetch_bytes_x2_gfx950"] = 2.0 * counters["FETCH_SIZE"] * 1024.0
        d["write_bytes"] = counters["WRITE_SIZE"] * 1024.0
        d["algorithmic_bytes_per_launch"] = ALGO_BYTES[wl]
        d["traffic_over_algorithmic"] = d["hbm_bytes_per_launch"] / ALGO_BYTES[wl]
    if "SQ_ACTIVE_INST_VALU" in counters and "GRBM_GUI_ACTIVE" in counters:
        # rocprof's VALUBusy = SQ_ACTIVE_INST_VALU * 4 / SIMDs / GRBM_GUI_ACTIVE; rocprofv3 reports GRBM_GUI_ACTIVE summed over
        # the 8 XCDs (MI355X_MICROARCH.md, DVFS note), so the per-XCD busy cycles are 1/8 of it.  256 CUs x 4 SIMDs.
        d["valu_busy"] = counters["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / (counters["GRBM_GUI_ACTIVE"] / 8.0)
        d["wait_inst_any_over_wave_cycles"] = counters.get("SQ_WAIT_INST_ANY", 0) / max(counters.get("SQ_WAVE_CYCLES", 1), 1)
        d["wait_any_over_wave_cycles"] = counters.get("SQ_WAIT_ANY", 0) / max(counters.get("SQ_WAVE_CYCLES", 1), 1)
    if "SQ_INSTS_VALU" in counters and meta:
        m_pairs = (10_000_000 if wl in ("k2", "pgmc", "vec1", "vec", "mixed") else ALGO_BYTES[wl] // 16) // 2
        d["valu_insts_per_pair_step"] = counters["SQ_INSTS_VALU"] * 64.0 / m_pairs / 64.0 * 64.0 / 64.0 * 1.0
        d["valu_insts_per_wave_iteration"] = counters["SQ_INSTS_VALU"] / (m_pairs / 64.0)
        for k in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
            if k in counters:
                d[k.lower().replace("sq_", "") + "_per_wave_iteration"] = counters[k] / (m_pairs / 64.0)
        if "SQ_LDS_BANK_CONFLICT" in counters and counters.get("SQ_INSTS_LDS"):
            d["lds_bank_conflict_cycles_per_lds_inst"] = counters["SQ_LDS_BANK_CONFLICT"] / counters["SQ_INSTS_LDS"]
    d.pop("valu_insts_per_pair_step", None)
    if entry.get("avg_us_kernel_trace"):
        d["achieved_GBps_algorithmic"] = ALGO_BYTES[wl] / entry["avg_us_kernel_trace"] / 1e3
        d["frac_of_8TBps"] = d["achieved_GBps_algorithmic"] / 8000.0
    entry["derived"] = d
    log = os.path.join(G, wl + ".log")
    if os.path.exists(log):
        entry["workload_line"] = [ln.strip() for ln in open(log) if ln.startswith(("ladder", "k2", "pgmc", "est", "vec", "mixed"))][-1:]
    summary[wl] = entry
for wl in ("vec2", "pgmc7"):          # kernel traces only (tools/gpu_profile.sh)
    f = one(os.path.join(G, wl, "trace", "*", "*_kernel_stats.csv"))
    if f:
        shutil.copy(f, os.path.join(OUT, f"{TAG}_{wl}_kernel_stats.csv"))
commit = os.environ.get("AMC_COMMIT") or subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
summary["_meta"] = dict(commit=commit, kernel_source_hash=kernel_hash(),
                        notes=["FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE counts 64 B per 128-B request on gfx950 (MI355X_MICROARCH.md, HBM): doubled",
                               "PMC passes are separate runs (counters only, 120 launches, first quarter dropped); durations come from the --kernel-trace --stats run",
                               "k2 / pgmc algorithmic bytes: 16 B (x read + write) + 1 B step-log byte per chain and step"])
json.dump(summary, open(os.path.join(OUT, f"{TAG}_pmc_summary.json"), "w"), indent=1)
head = summary.get("ladder_10000000", {}).get("derived", {})
if "hbm_bytes_per_launch" in head:
    json.dump({"sweep_kernel_bytes_per_launch": head["hbm_bytes_per_launch"], "sweep_kernel_valu_busy": head.get("valu_busy"),
               "source": f"profiles/{TAG}_pmc_summary.json", "commit": commit, "kernel_source_hash": kernel_hash()},
              open(os.path.join(OUT, "pmc_traffic.json"), "w"), indent=1)
for wl, e in summary.items():
    if wl.startswith("_"):
        continue
    d = e["derived"]
    print(f"{wl:18s} {e.get('avg_us_kernel_trace') or 0:8.2f} us  frac {d.get('frac_of_8TBps', 0):.3f}  traffic/algo {d.get('traffic_over_algorithmic', 0):.3f}  "
          f"VALUBusy {d.get('valu_busy', 0):.3f}  VALU/wave-iter {d.get('valu_insts_per_wave_iteration', 0):.1f}  "
          f"waitInst {d.get('wait_inst_any_over_wave_cycles', 0):.2f}  LDSconf/inst {d.get('lds_bank_conflict_cycles_per_lds_inst', 0):.1f}")
