#!/usr/bin/env python3
"""Developer tool (round 3): the VALU-issue floor of the three profiled kernels.

For each kernel: the instruction mix of ONE trip of its steady-state loop (from the ISA listing, `make -C montecarlo_amd/csrc asm`)
x the measured issue cost of every opcode (profiles/r03_ubench_issue_costs.txt, tools/ubench_issue.hip: ns per wave-instruction
per SIMD with 4 waves per SIMD) -> predicted microseconds per 1e7 chains if the vector unit did nothing but issue that stream,
next to the kernel-trace duration (profiles/<tag>_pmc_summary.json).  Writes profiles/r03_valu_floor.md.

How the loop trip is found: LLVM marks every basic block of a loop in the listing ("in Loop: Header=..." / "Parent Loop ...");
the steady-state loop is the one that holds the buffer loads of the next tile.  Inside it, the side of a conditional branch
that alone reaches a Philox call is a COLD arm -- the reference-ordered accept decision with its second Philox call and the
36-bit move-pick walk, taken by ~1.5 % / ~3 % of wave-steps -- which is weighted by its measured
frequency; everything else counts once per trip.  The resulting VALU count is checked against the SQ_INSTS_VALU counter.
"""
import collections, json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = os.environ.get("AMC_ROUND_TAG", "r04")
ASM = os.path.join(ROOT, "montecarlo_amd", "csrc", "amc_api.gfx950.s")
COSTS = os.path.join(ROOT, "profiles", "r03_ubench_issue_costs.txt")
if not os.path.exists(COSTS):
    COSTS = os.path.join(ROOT, "profiles", "r01_ubench_issue_costs.txt")
PMC = os.path.join(ROOT, "profiles", f"{TAG}_pmc_summary.json")
N_SIMD = 1024
LAUNCH_BOUNDARY_US = 2.9          # an empty launch back to back on this stack (tools/ubench_launch.hip)

KERNELS = [  # (title, demangled-name fragment, workload key in the PMC summary, chains per launch, cold-arm weight)
    ("K = 1 sweep (headline, config 2)", "sweep_kernel<0, false, 0, false, true, 0>", "ladder_10000000", 10_000_000, 0.015),
    ("K = 2 sweep (config 3)", "sweep_kernel<1, true, 1, false, true, 0>", "k2", 10_000_000, 0.045),
    ("fused PGMC time step (config 5)", "pg_estimate_kernel<0, 1, false, 2, 0, false>", "pgmc", 10_000_000, 0.045),
]


def load_costs():
    cost = {}
    for ln in open(COSTS):
        m = re.match(r"^(\S+)\s+([0-9.]+) ns/wave-instr/SIMD", ln)
        if m:
            cost[m.group(1)] = float(m.group(2))
    return cost


def opcode_cost(op, cost):
    """ns per wave-instruction per SIMD; measured where the microbenchmark has the opcode, by class otherwise."""
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base == "v_cndmask_b32":
        # the VOP2 form selects on vcc and issues at 22 cycles on this part (measured twice); the VOP3 form with the lane mask in
        # an SGPR pair -- what the hot loops' selects on ballots compile to -- at 4.5
        return (cost["v_cndmask_b32"], "measured") if op.endswith("_e32") else (cost.get("v_cndmask_b32_sgpr", cost["v_mov_b64"]), "measured")
    if base.startswith("v_cmp") and op.endswith("_e64"):
        # a compare into an SGPR pair (VOP3) costs more than the VOPC form that writes vcc
        return cost.get("v_cmp_lt_f64_sgpr" if "64" in base else "v_cmp_gt_f32_sgpr", cost["v_cmp_gt_f64"]), "measured"
    if base in cost:
        return cost[base], "measured"
    c = lambda k: cost[k]
    if base.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return cost.get("v_readlane_b32", c("v_mov_b32")), "class"
    if base in ("v_fmac_f64",):
        return c("v_fma_f64"), "class"
    if base in ("v_min_f64", "v_max_f64"):
        return c("v_max_f64"), "class"
    if base in ("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64"):
        return c(base), "measured"
    if base.startswith("v_cmp") and "f64" in base:
        return c("v_cmp_gt_f64"), "class"
    if base.startswith("v_cvt") and "f64" in base:
        return c("v_cvt_f64_u32"), "class"
    if "f64" in base or base.endswith("_b64") or base.endswith("_u64") or base.endswith("_i64"):
        return c("v_mov_b64"), "class"
    if base.startswith("v_mad_u64") or base.startswith("v_mad_i64"):
        return c("v_mad_u64_u32"), "class"
    if base.startswith(("v_mul_lo", "v_mul_hi")):
        return c("v_mul_lo_u32"), "class"
    if base in ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rsq_f32"):
        return c("v_exp_f32"), "class"
    if base.startswith("v_cvt"):
        return c("v_cvt_f32_u32"), "class"
    if base in ("v_alignbit_b32", "v_lshl_or_b32", "v_add3_u32", "v_and_or_b32", "v_lshl_add_u32", "v_add_lshl_u32", "v_perm_b32",
                "v_bfe_u32", "v_bfi_b32", "v_mad_u32_u24", "v_mul_u32_u24", "v_xad_u32", "v_or3_b32"):
        return c("v_alignbit_b32"), "class"          # three-operand 32-bit forms issue at the v_alignbit rate
    if base.startswith("v_cndmask"):
        return cost.get("v_cndmask_b32_sgpr", c("v_mov_b32")), "class"
    return c("v_add_u32"), "class"                   # plain 32-bit VALU


def demangle(name):
    return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()


def kernel_lines(asm, fragment):
    for name in re.findall(r"^(_ZN3amc[^:\s]*):", asm, re.M):
        if fragment in demangle(name):
            i = asm.index(name + ":")
            return asm[i:asm.index(".Lfunc_end", i)].splitlines()
    raise SystemExit(f"kernel {fragment} not in the listing")


def parse_blocks(lines):
    """[(label, loop_headers, [opcode, ...], [branch targets])] in layout order; a block ends at a label or after a branch."""
    blocks, cur = [], {"label": "entry", "loops": set(), "ops": [], "targets": []}
    for idx, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):(.*)$", l)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "loops": set(), "ops": [], "targets": []}
            note = m.group(2)
            j = idx
            while True:                                          # loop notes continue on comment-only lines
                for h in re.findall(r"(?:Header=|Parent Loop |Loop Header: )(BB\d+_\d+)?", note):
                    if h:
                        cur["loops"].add(".L" + h)
                if "Loop Header" in note:
                    cur["loops"].add(cur["label"])
                j += 1
                if j < len(lines) and lines[j].lstrip().startswith(";") and ("Loop" in lines[j]):
                    note = lines[j]
                else:
                    break
            continue
        t = l.strip()
        if not l.startswith("\t") or not t or t.startswith((".", ";")):
            continue
        op = t.split()[0]
        cur["ops"].append(op)
        if op.startswith(("s_cbranch", "s_branch")):
            cur["targets"].append(t.split()[1])
            nxt = {"label": cur["label"] + "+", "loops": set(cur["loops"]), "ops": [], "targets": []}
            blocks.append(cur)
            cur = nxt
    blocks.append(cur)
    # a block of an inner loop names only that loop; its header names the parents: close the membership over them
    by_label = {b["label"]: b for b in blocks}
    changed = True
    while changed:
        changed = False
        for b in blocks:
            for h in list(b["loops"]):
                extra = by_label[h]["loops"] - b["loops"] if h in by_label else set()
                if extra:
                    b["loops"] |= extra
                    changed = True
    return [b for b in blocks if b["ops"] or b["label"] != "entry"]


def steady_loop(blocks):
    """The loop (header label) whose blocks hold the most instructions among loops containing a buffer load."""
    size = collections.Counter()
    has_load = set()
    for b in blocks:
        for h in b["loops"]:
            size[h] += len(b["ops"])
            if any(o.startswith("buffer_load") for o in b["ops"]):
                has_load.add(h)
    cands = [h for h in size if h in has_load]
    return max(cands, key=lambda h: size[h])


def trip_mix(blocks, header, cold_weight):
    """Instruction mix of one loop trip.  Cold arms are found on the control-flow graph, not by layout: for a conditional
    branch with successors F (fall-through) and T (target), the blocks only one side reaches before the two sides meet again
    are that side's arm; an arm that holds a Philox call (>= 12 v_mad_u64_u32 outside inner loops) is one of the rare ones
    -- the kernels form a second Philox result only where the 12-bit brackets of the normal draw leave a decision open."""
    loop = [b for b in blocks if header in b["loops"]]
    idx = {b["label"]: i for i, b in enumerate(loop)}
    succ = []
    for i, b in enumerate(loop):
        out = [idx[t] for t in b["targets"] if t in idx and t != header]
        last = b["ops"][-1] if b["ops"] else ""
        if not last.startswith(("s_branch", "s_endpgm")) and i + 1 < len(loop):
            out.append(i + 1)
        succ.append(out)

    def reach(start):
        seen, todo = set(), [start]
        while todo:
            k = todo.pop()
            if k in seen:
                continue
            seen.add(k)
            todo.extend(succ[k])
        return seen

    inner = [len(b["loops"] - {header}) > 0 for b in loop]
    philox = lambda ks: sum(sum(1 for o in loop[k]["ops"] if "mad_u64" in o) for k in ks if not inner[k])
    weight = [1.0] * len(loop)
    arms = []
    for i, b in enumerate(loop):
        if not (b["ops"] and b["ops"][-1].startswith("s_cbranch")) or len(succ[i]) != 2:
            continue
        t, f = succ[i][0], succ[i][1]
        rt, rf = reach(t), reach(f)
        for side, other, name in ((rf - rt, rt - rf, "fall-through"), (rt - rf, rf - rt, "target")):
            # ... and so is the repair of alpha for NaN / arg < -708 in the estimator's sample (a v_cmp_u_f64 behind a
            # wave-uniform branch almost no wave takes, amc_kernels.h pg_sample)
            nan_repair = any("v_cmp_u_f64" in o for k in side for o in loop[k]["ops"]) and sum(len(loop[k]["ops"]) for k in side) < 40
            if (philox(side) >= 12 and philox(other) < 12) or nan_repair:
                for k in side:
                    weight[k] = min(weight[k], cold_weight)
                arms.append((b["label"], name, sum(len(loop[k]["ops"]) for k in side)))
    mix = collections.Counter()
    for b, w in zip(loop, weight):
        for op in b["ops"]:
            mix[op] += w
    return mix, arms, sum(len(b["ops"]) for b in loop)


def main():
    cost = load_costs()
    asm = open(ASM).read()
    pmc = json.load(open(PMC)) if os.path.exists(PMC) else {}
    out = []
    out.append(f"# VALU-issue floor of the profiled kernels ({TAG})\n")
    out.append("Per kernel: instruction mix of one trip of the steady-state loop (ISA listing of this commit; cold arms weighted by their "
               "measured frequency) x measured issue cost per opcode (`" + os.path.relpath(COSTS, ROOT) + "`, 4 waves per SIMD) = the time "
               "the vector unit needs to issue the stream, per 1e7 chains (78 125 wave-trips over 1024 SIMDs); + one launch boundary "
               f"({LAUNCH_BOUNDARY_US} us, an empty launch back to back on this stack); against the rocprofv3 kernel-trace average of the same "
               "commit (`profiles/" + TAG + "_pmc_summary.json`).  `tools/valu_floor.py` regenerates this file.\n")
    out.append("| kernel | VALU instr / wave-trip (ISA, weighted) | SQ_INSTS_VALU / wave-trip (PMC) | VALU issue, us / 1e7 chains | + launch boundary = floor | kernel-trace us | measured / floor | measured / VALU issue |")
    out.append("|---|---|---|---|---|---|---|---|")
    details = []
    for title, frag, wl, chains, cold_w in KERNELS:
        blocks = parse_blocks(kernel_lines(asm, frag))
        header = steady_loop(blocks)
        mix, arms, loop_instr = trip_mix(blocks, header, cold_w)
        valu = {op: n for op, n in mix.items() if op.startswith("v_")}
        n_valu = sum(valu.values())
        ns = 0.0
        by_class = collections.Counter()
        unmeasured = collections.Counter()
        for op, n in valu.items():
            c, how = opcode_cost(op, cost)
            ns += n * c
            cls = ("f64 arithmetic" if re.search(r"(fma|fmac|mul|add|min|max)_f64", op) else
                   "f64 rcp/rsq/sqrt" if re.search(r"(rcp|rsq|sqrt)_f64", op) else
                   "other 64-bit (cmp, cvt, mov, ldexp, shifts)" if re.search(r"f64|_b64|_u64$|_i64", op) and "mad_u64" not in op else
                   "v_mad_u64_u32 (Philox)" if "mad_u64" in op else
                   "f32 transcendental" if re.search(r"(exp|log|rcp|sqrt|sin|cos)_f32", op) else
                   "lane moves (readlane / writelane)" if "lane" in op else "32-bit")
            by_class[cls] += n
            by_class[cls + " ns"] += n * c
            if how == "class":
                unmeasured[op] += n
        trips_per_simd = chains / 2 / 64 / N_SIMD
        issue_us = ns * trips_per_simd / 1e3
        floor_us = issue_us + LAUNCH_BOUNDARY_US
        e = pmc.get(wl, {})
        meas = e.get("avg_us_kernel_trace")
        pmc_valu = (e.get("derived") or {}).get("valu_insts_per_wave_iteration")
        out.append(f"| {title}: `{frag}` | {n_valu:.0f} | {pmc_valu and f'{pmc_valu:.0f}' or 'n/a'} | {issue_us:.1f} | {floor_us:.1f} | "
                   f"{meas and f'{meas:.1f}' or 'n/a'} | {meas and f'{meas / floor_us:.2f}' or 'n/a'} | {meas and f'{meas / issue_us:.2f}' or 'n/a'} |")
        d = [f"\n## {title}\n", f"`{frag}`: loop header `{header}`, {loop_instr} instructions in the loop body "
             f"({sum(n for op, n in mix.items() if op.startswith('s_')):.0f} scalar, {sum(n for op, n in mix.items() if op.startswith('ds_')):.0f} LDS, "
             f"{sum(n for op, n in mix.items() if op.startswith(('buffer', 'global', 'flat'))):.0f} memory per trip, not priced: they issue beside "
             f"the vector stream); cold arms (weight {cold_w}): " + (", ".join(f"{n} instructions on the {b} side of the branch ending {a}" for a, b, n in arms) or "none") + ".\n",
             "| class | instructions / wave-trip | ns / wave-trip |", "|---|---|---|"]
        for cls in sorted({k for k in by_class if not k.endswith(" ns")}, key=lambda k: -by_class[k + " ns"]):
            d.append(f"| {cls} | {by_class[cls]:.1f} | {by_class[cls + ' ns']:.1f} |")
        d.append(f"| **total** | **{n_valu:.1f}** | **{ns:.1f}** |")
        if unmeasured:
            d.append("\nPriced by class (opcode not in the microbenchmark): " + ", ".join(f"{op} x{n:.0f}" for op, n in unmeasured.most_common(12)) + ".")
        details.extend(d)
    out.extend(details)
    text = "\n".join(out) + "\n"
    note = os.path.join(ROOT, "profiles", "r03_valu_floor_verdict.md")
    if os.path.exists(note):
        text += "\n" + open(note).read()
    open(os.path.join(ROOT, "profiles", TAG + "_valu_floor.md"), "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
